#!/usr/bin/env python3
"""Where the exact sum's cycles go (K1, RL_SUM_EXACT): runs one Paint of a synthetic N x L chunk on a library built
with -DRL_STATS (tools/build_paint_variant.sh stats "-DRL_STATS -DRL_ONLY_S=80", loaded through RELATE_AMD_LIB) and
prints the 16 event counters of exact_sum.h / paint_kernels.hip per sum.

    RELATE_AMD_LIB=$PWD/relate_amd/variants/librelate_amd_stats.so python tools/exp_stats.py [N L]

counters 0..7: forward sums, 8..15: backward sums -- number of sums, fallbacks to the literal order, lanes walked,
lanes re-run from their exact entry (two-binade jumps), cycles in the scan / the four chains / the classification and
map scan / the walk."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RELATE_AMD_STATS"] = "1"
import bench  # noqa: E402
from relate_amd import api  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    bits, r, rpos, wb = bench.make_chunk(N, L, 1, 20.0)
    ctx = api.Context()
    ctx.set_chunk_bits(N, bits, r, rpos, wb)
    out = {"N": N, "L": L, "W": len(wb) - 1, "total_sites": ctx.total_sites()}
    lib = api.lib()
    for split, name in ((1, "split"), (0, "merged")):
        lib.rl_set_paint_split(C.c_void_p(ctx._h), split)
        ms = ctx.paint(api.RL_SUM_EXACT)
        ms = ctx.paint(api.RL_SUM_EXACT)
        st = np.zeros(32, dtype=np.uint64)
        rc = lib.rl_debug_stats32(C.c_void_p(ctx._h), st.ctypes.data_as(C.c_void_p))
        out[name] = {"kernel_ms": ms, "rc": rc}
        if split:
            f, b = C.c_float(), C.c_float()
            lib.rl_paint_times(C.c_void_p(ctx._h), C.byref(f), C.byref(b))
            out[name]["fwd_ms"], out[name]["bwd_ms"] = f.value, b.value
        for d, o in (("forward", 0), ("backward", 8)):
            n = max(1, int(st[o]))
            out[name][d] = {"sums": int(st[o]), "fallbacks": int(st[o + 1]), "walked_lanes_per_sum": float(st[o + 2]) / n,
                            "rerun_lanes_per_sum": float(st[o + 3]) / n, "cycles_scan": float(st[o + 4]) / n,
                            "cycles_chains": float(st[o + 5]) / n, "cycles_classify": float(st[o + 6]) / n,
                            "cycles_walk": float(st[o + 7]) / n}
        nf, nb = max(1, int(st[0])), max(1, int(st[8]))
        out[name]["forward_step_cycles"] = {"whole": float(st[16]) / nf, "chunk_loop": float(st[17]) / nf,
                                            "sum": float(st[18]) / nf, "rest": float(st[19]) / nf}
        out[name]["backward_step_cycles"] = {"whole": float(st[20]) / nb, "divisions_slot_loads": float(st[21]) / nb,
                                             "chunk_loop": float(st[22]) / nb, "sum": float(st[23]) / nb,
                                             "rescale_factor": float(st[24]) / nb}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
