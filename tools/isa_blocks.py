#!/usr/bin/env python3
"""Basic-block instruction census of a gfx950 .s file (hipcc --save-temps): per kernel, every label-delimited block
with its count of vector-ALU, scalar, memory and branch instructions.  Used to see what the hot loops of the
painting kernels issue per step."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "vlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main(path, want=None, top=40):
    kern, blocks, cur = None, [], None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
            cur = [kern, "entry", Counter(), Counter()]
            blocks.append(cur)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m and kern:
            cur = [kern, m.group(1), Counter(), Counter()]
            blocks.append(cur)
            continue
        if line.startswith("\t.") or not line.startswith("\t") or cur is None:
            if line.startswith(".Lfunc_end"):
                kern, cur = None, None
            continue
        op = line.split()[0]
        if op.startswith(";"):
            continue
        cur[2][classify(op)] += 1
        cur[3][op] += 1
    for k in sorted(set(b[0] for b in blocks)):
        if want and want not in k:
            continue
        bs = [b for b in blocks if b[0] == k]
        print("==", k, "blocks", len(bs), "instr", sum(sum(b[2].values()) for b in bs))
        for b in sorted(bs, key=lambda b: -sum(b[2].values()))[:top]:
            tot = sum(b[2].values())
            if tot < 12:
                continue
            print("  %-12s %5d  %s   | %s" % (b[1], tot, dict(b[2]),
                                           " ".join("%s:%d" % x for x in b[3].most_common(8))))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 40)
