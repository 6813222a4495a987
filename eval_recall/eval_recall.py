"""Proposal-recall evaluation of WeDetect-Uni on the MI355X path — command-line compatible with the reference's
``eval_recall/eval_recall.py`` (flags 1499-1505, flow 1507-1589), launched the same way:

    torchrun --nproc_per_node 8 eval_recall/eval_recall.py --wedetect_uni_checkpoint wedetect_base_uni.pth --dataset coco

One process per GPU over RCCL; the image list is sharded contiguously per rank like the reference's ``InferenceSampler``
(1470-1488); every rank runs the Uni detector on real batches (the reference feeds the first image of each batch) and
keeps its <= 300 boxes per image on the host; the two ``all_gather_object`` calls of 1567-1571 stay as they are (ids and
[n, 4] box arrays: kilobytes per image); rank 0 builds the ground-truth lists exactly as ``fast_eval_recall`` does (27-135:
COCO drops ``ignore`` / ``iscrowd`` boxes, LVIS / PACO only ``ignore``) from the annotation JSON itself (pycocotools / lvis
are index helpers over that file) and prints AR@100 / AR@300 over IoU 0.5:0.05:0.95 from ``wedetect_amd.evaluate.eval_recalls``
(recall.py:118-178 with the IoU matrix and the greedy matching on the device).

``--ann-path`` / ``--image-path`` default to the reference's ``ds_collections`` for coco / lvis / paco."""
import argparse
import datetime
import itertools
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from generate_proposal import load_uni_detector  # noqa: E402
from wedetect_amd.evaluate import eval_recalls  # noqa: E402
from wedetect_amd.parallel import shard_range  # noqa: E402

DATASETS = {                                                          # eval_recall.py:10-23
    "coco": dict(ann_path="data/coco/annotations/instances_val2017.json", image_path="data/coco/val2017/"),
    "lvis": dict(ann_path="data/lvis/lvis_v1_val.json", image_path="data/coco/"),
    "paco": dict(ann_path="data/PACO/paco_lvis_v1_test.json", image_path="data/coco/"),
}


class ImageDataset(torch.utils.data.Dataset):
    """eval_recall.py:1421-1466: ``{'id', 'image'}`` per annotation-file image (``coco_url`` for LVIS-style files)."""

    def __init__(self, images, image_path: str, indices):
        self.images, self.image_path, self.indices = images, image_path, list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        from PIL import Image
        ann = self.images[self.indices[i]]
        name = ann["file_name"] if "file_name" in ann else ann["coco_url"].replace("http://images.cocodataset.org/", "")
        return {"id": int(ann["id"]), "image": Image.open(os.path.join(self.image_path, name)).convert("RGB")}


def ground_truth_boxes(annotations, image_ids, drop_crowd: bool):
    """The per-image ground-truth arrays of fast_eval_recall (eval_recall.py:46-66 / 75-105): xywh -> xyxy fp32, ``ignore``
    annotations dropped, ``iscrowd`` too for COCO; images without usable boxes get a [0, 4] array."""
    by_img = {}
    for ann in annotations:
        by_img.setdefault(int(ann["image_id"]), []).append(ann)
    out = []
    for iid in image_ids:
        boxes = []
        for ann in by_img.get(int(iid), []):
            if ann.get("ignore", False) or (drop_crowd and ann.get("iscrowd", 0)):
                continue
            x, y, w, h = ann["bbox"]
            boxes.append([x, y, x + w, y + h])
        out.append(np.asarray(boxes, dtype=np.float32) if boxes else np.zeros((0, 4)))
    return out


def fast_eval_recall(dataset: str, annotations, proposals):
    """eval_recall.py:27-135: AR@100 and AR@300, printed like the reference and returned."""
    iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
    preds = [np.asarray(p["boxes"], dtype=np.float32).reshape(-1, 4) for p in proposals]
    gts = ground_truth_boxes(annotations, [p["image_id"] for p in proposals], drop_crowd=dataset == "coco")
    recalls = eval_recalls(gts, preds, [100, 300], iou_thrs)
    ar100, ar300 = float(sum(recalls[0]) / len(recalls[0])), float(sum(recalls[1]) / len(recalls[1]))
    print(ar100)
    print(ar300)
    return ar100, ar300


def run(args):
    import torch.distributed as dist
    own_group = False
    local = int(os.getenv("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        # the communicator is bound to this rank's device up front (a barrier otherwise guesses it from the current context)
        kw = dict(device_id=torch.device("cuda", local)) if args.backend == "nccl" else {}
        dist.init_process_group(backend=args.backend, world_size=int(os.getenv("WORLD_SIZE", "1")), rank=int(os.getenv("RANK", "0")),
                                timeout=datetime.timedelta(seconds=float(os.getenv("WEDETECT_COLLECTIVE_TIMEOUT", "600"))), **kw)
        own_group = True
    rank, world = dist.get_rank(), dist.get_world_size()
    if "base" not in args.wedetect_uni_checkpoint and "large" not in args.wedetect_uni_checkpoint:
        raise NotImplementedError("Please name the ckpt properly")               # eval_recall.py:1514-1522
    model = load_uni_detector(args.wedetect_uni_checkpoint, num_prompts=256, precision=args.precision)
    cfg = DATASETS.get(args.dataset, {})
    ann_path = args.ann_path or cfg.get("ann_path")
    image_path = args.image_path if args.image_path is not None else cfg.get("image_path")
    if ann_path is None or image_path is None:
        raise SystemExit(f"unknown --dataset {args.dataset!r}: give --ann-path and --image-path")
    with open(ann_path) as f:
        ann_file = json.load(f)
    random.seed(args.seed)
    mine = shard_range(len(ann_file["images"]), world, rank)                     # InferenceSampler, 1470-1488
    loader = torch.utils.data.DataLoader(ImageDataset(ann_file["images"], image_path, mine), batch_size=args.batch_size,
                                         num_workers=args.num_workers, pin_memory=False, drop_last=False,
                                         collate_fn=lambda inputs: inputs, shuffle=False)
    try:
        from tqdm import tqdm
        it = tqdm(loader, disable=rank != 0)
    except ImportError:
        it = loader
    image_ids, all_boxes = [], []
    with torch.no_grad():
        for inputs in it:
            outputs = model([x["image"] for x in inputs])
            for x, o in zip(inputs, outputs):
                image_ids.append(x["id"])
                all_boxes.append(o["bboxes"].cpu())
    dist.barrier()
    merged_ids, merged_boxes = [None] * world, [None] * world
    dist.all_gather_object(merged_ids, image_ids)                                # eval_recall.py:1567-1571
    dist.all_gather_object(merged_boxes, all_boxes)
    merged_ids = list(itertools.chain.from_iterable(merged_ids))
    merged_boxes = list(itertools.chain.from_iterable(merged_boxes))
    result = None
    if rank == 0:
        print(f"Evaluating {args.dataset} ...")
        results = [{"image_id": int(i), "boxes": b} for i, b in zip(merged_ids, merged_boxes)]
        result = fast_eval_recall(args.dataset, ann_file["annotations"], results)
    dist.barrier()
    if own_group:
        dist.destroy_process_group()
    return result


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--wedetect_uni_checkpoint", type=str, default="")
    parser.add_argument("--dataset", type=str, default="")
    parser.add_argument("--batch-size", type=int, default=1)
    parser.add_argument("--num-workers", type=int, default=1)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--ann-path", type=str, default=None)
    parser.add_argument("--image-path", type=str, default=None)
    parser.add_argument("--precision", default=None, choices=["fp32", "fp16x3"])
    parser.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    return run(parser.parse_args(argv))


if __name__ == "__main__":
    main(sys.argv[1:])
