"""Per-class precision / recall / F1 of object retrieval — command-line compatible with the reference's
``eval_retrieval/retrieval_metric.py`` (``--model --dataset --thre``; scoring lines 362-377, report 379-396).

Reads ``{dataset}_{model}.pth`` written by extract_embedding.py, scores every image against every class on the device
(``wd_retrieval_max``: sigmoid(<e, t> * exp(scale) + bias), max over the image's regions — logits never materialised),
thresholds, and compares with the ground-truth image sets of the annotation file.  Ground truth (the reference builds
it with pycocotools / lvis at import time): image ids per category read straight from the COCO-format JSON;
``--class-names`` (JSON list) gives the names in bank-row order, default = the annotation file's categories sorted by id."""
import argparse
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from wedetect_amd.evaluate import (evaluate_retrieval_per_class, load_retrieval_file, macro_average,  # noqa: E402
                                   retrieval_predictions)

ANN = {"coco": "data/coco/annotations/instances_val2017.json", "lvis": "data/lvis/lvis_v1_minival_inserted_image_name.json"}


def ground_truth(ann_path: str):
    """(class names in category-id order, {name: set(image ids with at least one instance)})."""
    with open(ann_path) as f:
        ann = json.load(f)
    cats = sorted(ann["categories"], key=lambda c: c["id"])
    name_of = {c["id"]: c["name"] for c in cats}
    gt = defaultdict(set)
    for a in ann["annotations"]:
        gt[name_of[a["category_id"]]].add(int(a["image_id"]))
    names = [c["name"] for c in cats]
    return names, {n: gt.get(n, set()) for n in names}


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--model", type=str, default="")
    parser.add_argument("--dataset", type=str, default="")
    parser.add_argument("--thre", type=float, default=0.3)
    parser.add_argument("--ann-path", type=str, default=None)
    parser.add_argument("--class-names", type=str, default=None)
    parser.add_argument("--pred", type=str, default=None, help="retrieval file (default {dataset}_{model}.pth)")
    args = parser.parse_args(argv)
    names, gt = ground_truth(args.ann_path or ANN[args.dataset])
    if args.class_names:
        with open(args.class_names, encoding="utf-8") as f:
            names = json.load(f)
    pred = load_retrieval_file(args.pred or f"{args.dataset}_{args.model}.pth")
    predictions = retrieval_predictions(pred, names, args.thre)
    print("Starting evaluation...")
    results = evaluate_retrieval_per_class(predictions, gt)
    print("\n" + "=" * 80)
    print(f"{'Class':<15} {'Prec':<8} {'Recall':<8} {'F1':<8} {'Support':<8} {'Pred#':<8}")
    print("-" * 80)
    for cat in sorted(results, key=lambda x: results[x]["f1"], reverse=True):
        r = results[cat]
        print(f"{cat:<15} {r['precision']:<8} {r['recall']:<8} {r['f1']:<8} {r['support']:<8} {r['n_pred']:<8}")
    p, r, f1 = macro_average(results) if results else (0.0, 0.0, 0.0)
    print("-" * 80)
    print(f"{'Macro Avg':<15} {p:<8.4f} {r:<8.4f} {f1:<8.4f}")
    print("=" * 80)
    return results


if __name__ == "__main__":
    main(sys.argv[1:])
