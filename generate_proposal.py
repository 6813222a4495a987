"""Class-agnostic proposals + region embeddings (WeDetect-Uni) on the MI355X path — command-line compatible with the
reference's ``generate_proposal.py`` (flags and flow: generate_proposal.py:1222-1273):

    python generate_proposal.py --wedetect_uni_checkpoint wedetect_base_uni.pth --image demo.jpg \
        --score_thre 0.1 --num_proposals 300 [--visualize]

``from generate_proposal import SimpleYOLOWorldDetector`` keeps working for code written against the reference module
(the class is ``wedetect_amd.detector.SimpleYOLOWorldDetector``: same constructor, ``load_state_dict`` accepts the
checkpoint with or without the reference's key remap at 1236-1254, ``model([paths or PIL images])`` returns the same
list of dicts)."""
import argparse
import sys

import torch

from wedetect_amd.apis import load_checkpoint_file
from wedetect_amd.detector import SimpleYOLOWorldDetector, letterbox  # noqa: F401  (re-exported names)


def model_size_of(checkpoint_path: str) -> str:
    """generate_proposal.py:1231: the size is read off the file name."""
    return "base" if "base" in checkpoint_path else "large"


def load_uni_detector(checkpoint_path: str, num_prompts: int = 256, num_proposals: int = 300, precision=None, device=None):
    model = SimpleYOLOWorldDetector(backbone_size=model_size_of(checkpoint_path), prompt_dim=768, num_prompts=num_prompts,
                                    num_proposals=num_proposals, precision=precision)
    msg = model.load_state_dict(load_checkpoint_file(checkpoint_path), strict=False)
    print(msg)
    model = model.cuda(device)
    model.eval()
    return model


def plot_bounding_boxes(image, boxes, width: int = 2):
    from PIL import ImageDraw
    image = image.convert("RGB")
    draw = ImageDraw.Draw(image)
    for x1, y1, x2, y2 in boxes:
        draw.rectangle([x1, y1, x2, y2], outline=(255, 0, 0), width=width)
    return image


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--wedetect_uni_checkpoint", type=str, default="")
    parser.add_argument("--image", type=str, default="")
    parser.add_argument("--score_thre", type=float, default=0.1)
    parser.add_argument("--num_proposals", type=int, default=300)
    parser.add_argument("--visualize", action="store_true")
    parser.add_argument("--precision", default=None, choices=["fp32", "fp16x3"])
    parser.add_argument("--output", type=str, default="pred.png")
    args = parser.parse_args(argv)

    model = load_uni_detector(args.wedetect_uni_checkpoint, num_proposals=args.num_proposals, precision=args.precision)
    with torch.no_grad():
        outputs = model([args.image])
    pred_bboxes = outputs[0]["bboxes"].float().cpu()
    pred_scores = outputs[0]["scores"].float().cpu()
    if args.score_thre > 0:
        mask = pred_scores > args.score_thre
        pred_bboxes = pred_bboxes[mask]
        pred_scores = pred_scores[mask]
    print(f"{len(pred_scores)} proposals above {args.score_thre}")
    if args.visualize:
        from PIL import Image
        plot_bounding_boxes(Image.open(args.image), pred_bboxes.tolist()).save(args.output)
    return dict(bboxes=pred_bboxes, scores=pred_scores, embeddings=outputs[0]["embeddings"], outputs=outputs)


if __name__ == "__main__":
    main(sys.argv[1:])
