"""Region-embedding extraction for object retrieval on the MI355X path — command-line compatible with the reference's
``eval_retrieval/extract_embedding.py`` (flags 1655-1663, flow 1665-1775), launched the same way (README.md:122-141):

    torchrun --nproc_per_node 8 eval_retrieval/extract_embedding.py --model wedetect_base_uni \
        --wedetect_checkpoint wedetect_base.pth --wedetect_uni_checkpoint wedetect_base_uni.pth --dataset coco

One process per GPU over RCCL; the image list is sharded contiguously per rank exactly like the reference's
``InferenceSampler`` (1620-1644); each rank runs the Uni detector (letterbox, tower, top-k, NMS, embedding gather all
on the device, real batches instead of the reference's first-image-of-each-batch); the class names go through the
XLM-R text tower in chunks of 80 (1708-1713).  The exchange replaces the four pickled ``all_gather_object`` calls
(1753-1756) by a pipelined ``all_gather_into_tensor`` of each step's fixed-shape result block ([B, 300, 768]
embeddings + counts + scales + bias + ids, ``parallel.StreamedRecordCollector``): device memory stays at two staging
slots whatever the size of the image set, and only rank 0 keeps anything — the kept rows, on the host, as the
reference does; rank 0 writes ``{dataset}_{model}.pth`` in the reference's format
(1763-1774: ``{"image_embedding": [{image_id, embedding, scale, bias}], "text_embedding"}``), which
``retrieval_metric.py`` reads unchanged.

Where the data comes from (the reference hard-codes all of it in ``ds_collections``): ``--ann-path`` / ``--image-path``
default to the reference's locations for ``--dataset coco|lvis``; class prompts are read from ``--class-texts`` (a
JSON list of names or of name lists, first name used — the ``data/texts/*_class_texts.json`` files the configs use),
default ``data/texts/{dataset}_zh_class_texts.json`` (``lvis`` -> ``lvis_v1``).  ``--text-bank`` replaces the text
tower by a precomputed ``[K, 768]`` bank."""
import argparse
import datetime
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
from wedetect_amd.apis import load_checkpoint_file  # noqa: E402
from wedetect_amd.evaluate import retrieval_records, save_retrieval_file  # noqa: E402
from wedetect_amd.parallel import StreamedRecordCollector, shard_range  # noqa: E402
from wedetect_amd.text import XLMRobertaLanguageBackbone  # noqa: E402

DATASETS = {
    "coco": dict(ann_path="data/coco/annotations/instances_val2017.json", image_path="data/coco/val2017/",
                 class_texts="data/texts/coco_zh_class_texts.json", key="file_name"),
    "lvis": dict(ann_path="data/lvis/lvis_v1_minival_inserted_image_name.json", image_path="data/coco/",
                 class_texts="data/texts/lvis_v1_zh_class_texts.json", key="coco_url"),
}
TEXT_CHUNK = 80                                             # extract_embedding.py:1708-1712


class ImageDataset(torch.utils.data.Dataset):
    """extract_embedding.py:1576-1617: ``{'id', 'image'}`` per annotation-file image, decoded to RGB."""

    def __init__(self, ann_path: str, image_path: str, key: str = "file_name", indices=None):
        with open(ann_path) as f:
            images = json.load(f)["images"]
        self.images = []
        for ann in images:
            name = ann[key] if key in ann else ann["file_name"]
            name = name.replace("http://images.cocodataset.org/", "")
            self.images.append({"id": ann["id"], "image": os.path.join(image_path, name)})
        self.indices = list(indices) if indices is not None else list(range(len(self.images)))

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        from PIL import Image
        ann = self.images[self.indices[i]]
        return {"id": int(ann["id"]), "image": Image.open(ann["image"]).convert("RGB")}


def collate_fn(inputs):
    return inputs


def text_encoder_from_checkpoint(ckpt_path: str, tokenizer=None, precision=None) -> XLMRobertaLanguageBackbone:
    """extract_embedding.py:1267-1304: size and tokenizer directory from the checkpoint's file name, weights =
    its ``backbone.text_model.*`` tensors."""
    if "base" in ckpt_path:
        size, name = "base", "../xlm-roberta-base/"
    elif "large" in ckpt_path:
        size, name = "large", "../xlm-roberta-large/"
    else:
        raise NotImplementedError("Please name the ckpt properly (base / large)")
    enc = XLMRobertaLanguageBackbone(model_name=name, model_size=size, tokenizer=tokenizer, precision=precision)
    sd = load_checkpoint_file(ckpt_path)
    enc.load_state_dict({k: v for k, v in sd.items() if k.startswith("backbone.text_model.")})
    return enc.cuda()


def encode_class_names(encoder, names, chunk: int = TEXT_CHUNK) -> torch.Tensor:
    """[K, 768] unit rows: chunks of 80 names through the tower (1708-1713; the tower's last step is the normalise)."""
    rows = [encoder.encode_classes(names[i:i + chunk]) for i in range(0, len(names), chunk)]
    return torch.cat(rows) if rows else torch.empty(0, 768, device="cuda")


def read_class_names(path: str):
    with open(path, encoding="utf-8") as f:
        data = json.load(f)
    return [(c[0] if isinstance(c, (list, tuple)) else c) for c in data]


def run(args, tokenizer=None):
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
    # the closing barrier sits behind rank 0 writing a multi-GB file: it gets a group of its own with a long deadline
    # ($WEDETECT_SAVE_TIMEOUT, default 2 h) instead of the 600 s every data-path collective is held to
    save_group = dist.new_group(timeout=datetime.timedelta(seconds=float(os.getenv("WEDETECT_SAVE_TIMEOUT", "7200")))) if world > 1 else None
    dev = torch.device("cuda", torch.cuda.current_device())
    if "base" not in args.wedetect_uni_checkpoint and "large" not in args.wedetect_uni_checkpoint:
        raise NotImplementedError("Please name the ckpt properly")                    # extract_embedding.py:1679-1681
    model = load_uni_detector(args.wedetect_uni_checkpoint, num_prompts=256, precision=args.precision)

    ds_cfg = dict(DATASETS.get(args.dataset, {}))
    ann_path = args.ann_path or ds_cfg.get("ann_path")
    image_path = args.image_path if args.image_path is not None else ds_cfg.get("image_path")
    if ann_path is None or image_path is None:
        raise SystemExit(f"unknown --dataset {args.dataset!r}: give --ann-path and --image-path")
    if args.text_bank:
        bank = np.load(args.text_bank) if args.text_bank.endswith(".npy") else torch.load(args.text_bank, map_location="cpu")
        text_embeddings = torch.as_tensor(np.asarray(bank), dtype=torch.float32).to(dev)
    else:
        names = read_class_names(args.class_texts or ds_cfg.get("class_texts"))
        encoder = text_encoder_from_checkpoint(args.wedetect_checkpoint, tokenizer, args.precision)
        with torch.no_grad():
            text_embeddings = encode_class_names(encoder, names)

    random.seed(args.seed)
    full = ImageDataset(ann_path, image_path, ds_cfg.get("key", "file_name"))
    mine = shard_range(len(full.images), world, rank)                                 # InferenceSampler, 1631-1638
    dataset = ImageDataset(ann_path, image_path, ds_cfg.get("key", "file_name"), indices=mine)
    loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, num_workers=args.num_workers,
                                         pin_memory=False, drop_last=False, collate_fn=collate_fn, shuffle=False)
    r = model.num_proposals
    # every rank runs the same number of steps (the first ranks of the split hold one image more): ranks whose shard
    # is exhausted contribute count-0 blocks
    n_steps = -(-len(shard_range(len(full.images), world, 0)) // args.batch_size)
    collector = StreamedRecordCollector(args.batch_size, r, 768, dev)
    try:
        from tqdm import tqdm
        it = tqdm(loader, disable=rank != 0)
    except ImportError:
        it = loader
    steps = 0
    with torch.no_grad():
        for inputs in it:
            res, counts, tower = model.forward_batch([x["image"] for x in inputs])
            ls = torch.tensor(tower.lvl_logit_scale, dtype=torch.float32, device=dev)
            cb = torch.tensor(tower.lvl_bias, dtype=torch.float32, device=dev)
            lvl = tower.level_of(res["anchors"])
            collector.step(res["embeddings"], res["count"], ls[lvl], cb[lvl],
                           torch.tensor([x["id"] for x in inputs], dtype=torch.int64, device=dev))
            steps += 1
        for _ in range(steps, n_steps):
            collector.pad_step()
    records = collector.finish()
    dist.barrier()
    out_path = args.output or f"{args.dataset}_{args.model}.pth"
    if rank == 0:
        print(f"Evaluating {args.dataset} ...")
        save_retrieval_file(out_path, records, text_embeddings)
        print(f"wrote {out_path}: {len(records)} images, {tuple(text_embeddings.shape)} text bank")
    if save_group is not None:
        dist.barrier(group=save_group)
    if own_group:
        dist.destroy_process_group()
    return out_path


def main(argv=None, tokenizer=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--model", type=str, default="")
    parser.add_argument("--wedetect_checkpoint", type=str, default="")
    parser.add_argument("--wedetect_uni_checkpoint", type=str, default="")
    parser.add_argument("--dataset", type=str, default="")
    parser.add_argument("--batch-size", type=int, default=1)
    parser.add_argument("--num-workers", type=int, default=1)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--ann-path", type=str, default=None)
    parser.add_argument("--image-path", type=str, default=None)
    parser.add_argument("--class-texts", type=str, default=None)
    parser.add_argument("--text-bank", type=str, default=None)
    parser.add_argument("--output", type=str, default=None)
    parser.add_argument("--precision", default=None, choices=["fp32", "fp16x3"])
    parser.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    return run(parser.parse_args(argv), tokenizer)


if __name__ == "__main__":
    main(sys.argv[1:])
