"""gfx9 hazard "VALU writes SGPR -> VMEM reads that SGPR needs 5 wait states", for VMEM instructions issued from INLINE ASM
(;;#ASMSTART ... ;;#ASMEND blocks): the compiler's hazard recognizer does not look inside them, and it does place VALU writes of
SGPRs (v_readlane_b32 restores of spilled SGPRs, v_readfirstlane_b32) directly in front of such a block.  For every VMEM
instruction inside an asm block, walks back over the preceding instructions (each = 1 wait state, s_nop N = N + 1) and
reports a VALU instruction writing one of its SGPR operands less than 5 wait states ahead.

    python scripts/check_sgpr_vmem_hazard.py file.s [file.s ...]        exit code 1 on any hazard
"""
import re
import sys


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    lines = open(path).read().split("\n")
    ins = []            # (line number, text, in_asm)
    in_asm = False
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        c = l.split(";")[0].strip()
        if not c or c.endswith(":") or c.startswith("."):
            continue
        ins.append((i + 1, c, in_asm))
    bad = 0
    for k, (ln, c, a) in enumerate(ins):
        toks = re.split(r"[ ,]+", c)
        if not a or not toks[0].startswith(("global_", "buffer_", "flat_", "scratch_")):
            continue
        need = set()
        for t in toks[1:]:
            need |= sregs(t)
        if not need:
            continue
        ws, j = 0, k - 1
        while j >= 0 and ws < 5:
            pl, pc, _ = ins[j]
            pt = re.split(r"[ ,]+", pc)
            if pt[0].startswith("v_") and len(pt) > 1 and (sregs(pt[1]) & need):
                print(f"{path}:{ln}: '{c}' reads s{sorted(sregs(pt[1]) & need)} written by VALU '{pc}' (line {pl}) only {ws} wait state(s) earlier")
                bad += 1
                break
            ws += int(pt[1], 0) + 1 if pt[0] == "s_nop" else 1
            j -= 1
    return bad


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        b = check(p)
        print(f"{p}: {b} hazard(s)")
        total += b
    sys.exit(1 if total else 0)
