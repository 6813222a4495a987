#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (rocpd sqlite outputs):

    python scripts/traffic_json.py <pmc_fetch.db> <pmc_write.db> > profiles/rNN_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports
half of the bytes of 16 B/lane coalesced reads — confirmed here on the LayerNorm kernel whose byte
count is known)."""
import json
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    tot, disp = defaultdict(float), defaultdict(set)
    for name, d, ctr, val in cur.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        if ctr == counter:
            tot[short(name)] += val
            disp[short(name)].add(d)
    return {k: (tot[k], len(disp[k])) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from wedetect_amd.build import source_hash
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --steps 2 --warmup 1`; "
                   "KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of 16 B/lane coalesced reads)",
           # provenance: bench.py prints these beside roofline.traffic and says whether the running library was built from the same sources
           "commit": os.environ.get("WD_COMMIT"), "kernel_source_sha256": source_hash(),
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        out["kernels"][k] = {"launches": n, "fetch_kib_raw_per_launch": f / max(nf, 1), "write_kib_per_launch": w / max(nw, 1),
                             "hbm_bytes_per_launch": 1024.0 * (2.0 * f / max(nf, 1) + w / max(nw, 1))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
