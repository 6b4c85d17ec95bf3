"""experiment: Paint as one launch (backward blocks first / interleaved), two launches on one stream, two streams"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
import bench
N, L, mem = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
bits, r, rpos, wb = bench.make_chunk(N, L, 1, mem)
ctx = api.Context(0); ctx.set_chunk_bits(N, bits, r, rpos, wb)
for mode, name in ((api.RL_SUM_EXACT, "exact"), (api.RL_SUM_LANES, "lanes")):
    ctx.paint(mode)
    for split, order, what in ((1, "0", "two launches"), (0, "0", "one launch, backward first"),
                               (0, "1", "one launch, interleaved"), (2, "0", "two streams")):
        os.environ["RELATE_AMD_PAINT_ORDER"] = order
        ctx.set_paint_split(split)
        ms = ctx.paint(mode)
        print(name, what, "%.1f ms" % ms, ctx.paint_times(), flush=True)
