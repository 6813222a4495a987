"""Static check of a kernel's ISA for the inline-asm load idiom of split_gemm_mlpw.hip / stem.hip: a register that is the
destination of a global_load issued from inline asm must not be READ OR WRITTEN by any instruction until an s_waitcnt vmcnt(N)
has retired that load (N counted over the VMEM operations issued after it — loads return in order).  The compiler does not
know such a register is pending: if nothing ties it into a later wait it may hand it to another value (observed: the
epilogue's address arithmetic -> a memory fault).  Linear scan plus one extra pass over every loop (backward branch) with the
pending set of the branch point, so loads that are carried round a loop are followed into its top.

    python scripts/check_asm_loads.py <file.s> <kernel symbol substring> [...]      exit code 1 on any hazard
"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(body, start, end, pending, report):
    """simulate body[start:end); pending = list of [line, regs, vmem ops issued after]; returns the pending list at `end`"""
    pending = [list(p) for p in pending]
    bad = 0
    for i in range(start, end):
        l = body[i].split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        toks = re.split(r"[ ,]+", l)
        op = toks[0]
        if op.startswith("s_waitcnt") and "vmcnt" in l:
            n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
            pending = [p for p in pending if p[2] < n]
            continue
        allregs = set()
        for t in toks[1:]:
            allregs |= regs(t)
        for p in pending:
            if allregs & p[1]:
                if report:
                    print(f"  HAZARD line {i}: '{l}' touches v{sorted(allregs & p[1])} while the load of line {p[0]} is in flight")
                bad += 1
        if op.startswith(("global_load", "global_store", "buffer_", "scratch_", "flat_")):
            for p in pending:
                p[2] += 1
            if op.startswith("global_load") and "lds" not in op:
                pending.append([i, regs(toks[1]), 0])
    return pending, bad


def check(lines, sym):
    sts = [i for i, l in enumerate(lines) if l.endswith(":") and sym in l and not l.startswith((".", " ", "\t"))]
    if not sts:
        print(f"{sym}: not found")
        return 1
    return sum(check_one(lines, st, lines[st][:-1]) for st in sts)


def check_one(lines, st, sym):
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st:en]
    labels = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
    _, bad = scan(body, 0, len(body), [], True)
    # loops: a branch to a label above it; re-enter the loop body with what was pending at the branch
    for i, l in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+\s+(\S+)", l) or re.match(r"\s*s_branch\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            pend, _ = scan(body, 0, i, [], False)
            pend = [[p[0] - 10 ** 6, p[1], p[2]] for p in pend]           # mark as carried round the loop
            _, b2 = scan(body, labels[m.group(1)], i, pend, True)
            bad += b2
    print(f"{sym}: {len(body)} instructions, {bad} hazard(s)")
    return bad


if __name__ == "__main__":
    lines = [l.split(";")[0].rstrip() for l in open(sys.argv[1]).read().split("\n")]
    total = sum(check(lines, s) for s in sys.argv[2:])
    sys.exit(1 if total else 0)
