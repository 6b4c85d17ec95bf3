#!/usr/bin/env python3
"""K1 timing of one library build: RELATE_AMD_LIB=<variant .so> python tools/exp_paint_time.py MODE [N L]
-> one JSON line {lib, mode, kernel_ms (both directions in one launch), fwd_ms, bwd_ms (one direction alone)}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relate_amd import api  # noqa: E402

mode = {"exact": 0, "lanes": 1, "serial": 2, "lanes32": 3}[sys.argv[1] if len(sys.argv) > 1 else "lanes32"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
bits, r, rpos, wb = bench.make_chunk(N, L, 1, 20.0)
ctx = api.Context()
ctx.set_chunk_bits(N, bits, r, rpos, wb)
ctx.prepare()
ctx.paint(mode)
ms = min(ctx.paint(mode) for _ in range(3))
ctx.set_paint_split(True)
ctx.paint(mode)
f, b = ctx.paint_times()
print(json.dumps({"lib": os.path.basename(os.environ.get("RELATE_AMD_LIB", "default")), "mode": sys.argv[1] if len(sys.argv) > 1 else "lanes32",
                  "N": N, "L": L, "kernel_ms": round(ms, 2), "fwd_ms": round(f, 2), "bwd_ms": round(b, 2)}), flush=True)
