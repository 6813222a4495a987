"""Static check of a kernel's ISA for ds_reads issued from inline asm (split_conv3w_kernel, free-running form): the destination
registers of a ds_read must not be READ OR WRITTEN by any instruction until an s_waitcnt lgkmcnt(N) has retired it (LDS
operations return in order: N counts the DS operations issued after it; with a scalar load in flight only lgkmcnt(0) retires
anything).  The compiler tracks its own ds_reads and cannot see the asm ones: a register copy it places between an asm read
and the kernel's explicit wait would read stale data.  The rule is checked for EVERY ds_read of the kernel (the compiler's own
satisfy it by construction).  Linear scan plus one pass round every loop, as scripts/check_asm_loads.py.

    python scripts/check_asm_ds_reads.py <file.s> <kernel symbol substring> [...]      exit code 1 on any hazard
"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(body, start, end, pending, smem, report):
    """pending = list of [line, regs, DS ops issued after]; smem = a scalar load may be in flight"""
    pending = [list(p) for p in pending]
    bad = 0
    for i in range(start, end):
        l = body[i].split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        toks = re.split(r"[ ,]+", l)
        op = toks[0]
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending, smem = [], False
                elif not smem:
                    pending = [p for p in pending if p[2] < n]
            continue
        allregs = set()
        for t in toks[1:]:
            allregs |= regs(t)
        for p in pending:
            if allregs & p[1]:
                if report:
                    print(f"  HAZARD line {i}: '{l}' touches v{sorted(allregs & p[1])} while the ds_read of line {p[0]} is in flight")
                bad += 1
        if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            smem = True
        if op.startswith("ds_"):
            for p in pending:
                p[2] += 1
            if op.startswith("ds_read"):
                pending.append([i, regs(toks[1]), 0])
    return pending, smem, bad


def check_one(lines, st, sym):
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st:en]
    labels = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
    _, _, bad = scan(body, 0, len(body), [], False, True)
    for i, l in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+\s+(\S+)", l) or re.match(r"\s*s_branch\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            pend, smem, _ = scan(body, 0, i, [], False, False)
            pend = [[p[0] - 10 ** 6, p[1], p[2]] for p in pend]
            _, _, b2 = scan(body, labels[m.group(1)], i, pend, smem, True)
            bad += b2
    print(f"{sym}: {len(body)} instructions, {bad} ds_read hazard(s)")
    return bad


def check(lines, sym):
    sts = [i for i, l in enumerate(lines) if l.endswith(":") and sym in l and not l.startswith((".", " ", "\t"))]
    if not sts:
        print(f"{sym}: not found")
        return 1
    return sum(check_one(lines, st, lines[st][:-1]) for st in sts)


if __name__ == "__main__":
    lines = [l.split(";")[0].rstrip() for l in open(sys.argv[1]).read().split("\n")]
    total = sum(check(lines, s) for s in sys.argv[2:])
    sys.exit(1 if total else 0)
