"""Open-vocabulary detection demo on the MI355X path — command-line compatible with the reference's
``infer_wedetect.py`` (its flags: infer_wedetect.py:59-98; its flow: 150-195):

    python infer_wedetect.py --config config/wedetect_base.py --checkpoint ckpt.pth \
        --image demo.jpg --text '鞋,床' --threshold 0.3 --device cuda:0

config -> ``init_detector`` -> ``Compose(cfg.test_pipeline)`` -> texts ``[[t], ...] + [[' ']]`` (the blank class the
reference appends) -> ``model.reparameterize(texts)`` -> per image: pipeline, ``test_step``, score filter, top-k,
annotated copy in ``--output-dir``.  Everything numeric runs through libwedetect_hip.so.

Additions (absent from the reference, all optional): ``--text-bank FILE`` loads a precomputed ``[K + 1, 768]`` class
bank (``.npy`` / ``.pt``; rows for the K prompts then the blank) instead of running the text tower — for hosts without
the XLM-R tokenizer files; ``--precision {fp32,fp16x3}``; ``--dump-json`` also writes the kept detections per image.
"""
import argparse
import json
import os
import os.path as osp
import random
import sys

import numpy as np
import torch

from wedetect_amd.apis import inference_detector, init_detector
from wedetect_amd.cfgfile import Config, DictAction
from wedetect_amd.pipeline import Compose


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Demo")
    parser.add_argument("--config", help="test config file path")
    parser.add_argument("--checkpoint", help="checkpoint file")
    parser.add_argument("--image", help="image path, include image file or dir.")
    parser.add_argument("--text", help="text prompts, including categories separated by a comma or a txt file with "
                                       "each line as a prompt.")
    parser.add_argument("--topk", default=100, type=int, help="keep topk predictions.")
    parser.add_argument("--threshold", default=0.05, type=float, help="confidence score threshold for predictions.")
    parser.add_argument("--device", default="cuda:0", help="device used for inference.")
    parser.add_argument("--show", action="store_true", help="show the detection results.")
    parser.add_argument("--amp", action="store_true", help="accepted for compatibility: the device path always "
                                                           "computes at fp32-equivalent accuracy")
    parser.add_argument("--output-dir", default="demo_outputs", help="the directory to save outputs")
    parser.add_argument("--cfg-options", nargs="+", action=DictAction,
                        help="override some settings in the used config, the key-value pair in xxx=yyy format will be "
                             "merged into config file.")
    parser.add_argument("--text-bank", default=None, help="precomputed [K+1, 768] class embeddings (.npy / .pt)")
    parser.add_argument("--precision", default=None, choices=["fp32", "fp16x3"])
    parser.add_argument("--dump-json", action="store_true", help="write <image>.json with the kept detections")
    return parser.parse_args(argv)


def read_texts(spec: str):
    """infer_wedetect.py:162-167."""
    if spec.endswith(".txt"):
        with open(spec) as f:
            lines = f.readlines()
        return [[t.rstrip("\r\n")] for t in lines] + [[" "]]
    return [[t.strip()] for t in spec.split(",")] + [[" "]]


def list_images(path: str):
    """infer_wedetect.py:174-180."""
    if not osp.isfile(path):
        return [osp.join(path, img) for img in sorted(os.listdir(path)) if img.endswith(".png") or img.endswith(".jpg")]
    return [path]


def load_bank(path: str) -> torch.Tensor:
    bank = np.load(path) if path.endswith(".npy") else torch.load(path, map_location="cpu")
    if isinstance(bank, dict):
        bank = bank.get("text_embedding", next(iter(bank.values())))
    return torch.as_tensor(np.asarray(bank), dtype=torch.float32)


def visualize(output_file, image_path, bboxes, labels):
    from PIL import Image, ImageDraw, ImageFont
    image = Image.open(image_path).convert("RGB")
    draw = ImageDraw.Draw(image)
    try:
        font = ImageFont.truetype("simsun.ttc", 20)          # the reference's CJK font, when the host has it
    except OSError:
        font = ImageFont.load_default()
    rnd = random.Random(0)
    for box, label in zip(bboxes, labels):
        color = (rnd.randint(0, 255), rnd.randint(0, 255), rnd.randint(0, 255))
        x1, y1, x2, y2 = (float(v) for v in box)
        draw.rectangle([x1, y1, x2, y2], outline=color, width=2)
        draw.rectangle([x1, y1, x1 + len(label) * 20, y1 + 20], fill=color)
        try:
            draw.text((x1 + 2, y1 + 2), label, font=font, fill="white")
        except UnicodeEncodeError:                           # bitmap fallback font without CJK glyphs
            draw.text((x1 + 2, y1 + 2), label.encode("ascii", "replace").decode(), font=font, fill="white")
    image.save(output_file)


def main(argv=None, tokenizer=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    cfg.work_dir = osp.join("./work_dirs", osp.splitext(osp.basename(args.config))[0])
    model = init_detector(cfg, checkpoint=args.checkpoint, device=args.device, palette=["red"], tokenizer=tokenizer,
                          precision=args.precision)
    test_pipeline = Compose(cfg.test_pipeline)
    texts = read_texts(args.text)
    if not osp.exists(args.output_dir):
        os.makedirs(args.output_dir)
    images = list_images(args.image)
    if args.text_bank:
        bank = load_bank(args.text_bank)
        if bank.shape != (len(texts), 768):
            raise SystemExit(f"--text-bank holds {tuple(bank.shape)}, expected ({len(texts)}, 768): one row per prompt "
                             f"plus the blank class")
        model.set_text_embeddings(bank, texts)
    else:
        model.reparameterize(texts)
    results = []
    for n, image_path in enumerate(images):
        pred = inference_detector(model, image_path, texts, test_pipeline, args.topk, args.threshold)
        labels = [f"{texts[c][0]} {s:0.2f}" for c, s in zip(pred["labels"], pred["scores"])]
        out = osp.join(args.output_dir, osp.basename(image_path))
        visualize(out, image_path, pred["bboxes"], labels)
        if args.dump_json:
            with open(osp.splitext(out)[0] + ".json", "w") as f:
                json.dump(dict(image=image_path, bboxes=pred["bboxes"].tolist(), scores=pred["scores"].tolist(),
                               labels=pred["labels"].tolist(), texts=[t[0] for t in texts]), f, ensure_ascii=False)
        results.append(pred)
        print(f"[{n + 1}/{len(images)}] {image_path}: {len(pred['scores'])} detections -> {out}", flush=True)
    return results


if __name__ == "__main__":
    main(sys.argv[1:])
