#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the text files kept under profiles/.

    python scripts/rocpd_summary.py <trace.db> [--pmc <pmc.db> ...] > profiles/rNN_summary.txt
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    print(f"## kernel trace: {path}")
    print(f"{'kernel':<58}{'calls':>7}{'total_ms':>11}{'avg_us':>10}{'min_us':>10}{'max_us':>10}{'pct':>7}")
    for n, c, s, a, mn, mx in rows:
        print(f"{short(n)[:57]:<58}{c:>7}{s/1e6:>11.3f}{a/1e3:>10.2f}{mn/1e3:>10.2f}{mx/1e3:>10.2f}{100*s/tot:>7.2f}")
    print(f"{'TOTAL':<58}{sum(r[1] for r in rows):>7}{tot/1e6:>11.3f}")


def pmc_stats(path):
    cur = sqlite3.connect(path).cursor()
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    dur = defaultdict(float)
    seen = set()
    for name, disp, ctr, val, d in cur.execute(
            "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
        k = short(name)
        acc[k][ctr] += val
        calls[k].add(disp)
        if (disp, ) not in seen:
            seen.add((disp, ))
            dur[k] += d
    ctrs = sorted({c for v in acc.values() for c in v})
    print(f"\n## PMC sums per kernel (all dispatches): {path}")
    print(f"{'kernel':<44}{'calls':>6}" + "".join(f"{c[-22:]:>24}" for c in ctrs))
    for k in sorted(acc, key=lambda k: -dur[k]):
        print(f"{k[:43]:<44}{len(calls[k]):>6}" + "".join(f"{acc[k].get(c, 0.0):>24.4g}" for c in ctrs))
    return acc, calls


if __name__ == "__main__":
    args = sys.argv[1:]
    kernel_stats(args[0])
    for p in args[1:]:
        if p != "--pmc":
            pmc_stats(p)
