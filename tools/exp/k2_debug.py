import os, sys, numpy as np, tempfile, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from test_golden_gpu import Fixture, open_ctx, u32
from relate_amd import api
tmp = pathlib.Path(tempfile.mkdtemp())
fx = Fixture("synth24", tmp, None)
ctx = open_ctx(fx)
pdir = str(tmp / "refpaint")
fx.write_paint_files(pdir)
for w in fx.dump_windows():
    s0 = int(fx.chunk.wb[w])
    win = ctx.open_window(w, os.path.join(pdir, "relate_%d.bin" % w), s0, api.RL_SUM_EXACT)
    for n, (ls, top) in enumerate(fx.repaint(w)):
        gtop, gls = win.topology(n)
        bad = [j for j in range(len(ls)) if not np.array_equal(u32(gtop[j]), u32(top[j]))]
        print("window", w, "target", n, "rows", len(ls), "bad rows", bad[:40], "ls ok", np.array_equal(u32(gls), u32(ls)))
        if bad and n < 2:
            j = bad[0]
            print("  row", j, "got", gtop[j][:8], "exp", top[j][:8])
    win.close()
    break
