#!/usr/bin/env python3
"""BASELINE.json config #5's N through config #5's ROUTE, on what this environment has -- one GPU: a synthetic
N = 10,000 chunk (L = 20,000 by default: a tenth of C5's 200k SNPs; the whole of C5 holds 288 GB of stepping stones and
needs the eight GPUs) through relate_amd.dist.run_chunk_by_targets -- `ranks` target ranges as threads on the GPU
(1 by default), every matrix assembled from all ranks' rows, the trees built by the device workers.

    python tools/chunk_c5_sharded.py [N L memory_GB sections in_flight window_fraction ranks]
    (window_fraction: share of a window's posterior rows kept resident; 1: all; < 0: from the free HBM, the default of
    run_chunk_by_targets)

Prints one JSON line: wall-clock of Paint + BuildTopology of the first `sections` sections, trees, trees/s, md5 of
section 0's files (tools/chunk_wallclock_big.py N L mem 1 ref gives the reference's for the same chunk)."""
import ctypes as C, hashlib, json, os, shutil, struct, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api, dist as rdist

a = sys.argv[1:]
N = int(a[0]) if len(a) > 0 else 10000
L = int(a[1]) if len(a) > 1 else 20000
mem = float(a[2]) if len(a) > 2 else 25.0
sections = int(a[3]) if len(a) > 3 else 8
in_flight = int(a[4]) if len(a) > 4 else 8
frac = float(a[5]) if len(a) > 5 else 0.25
ranks = int(a[6]) if len(a) > 6 else 1
lib = api.lib()
seq = np.zeros((L, N), dtype=np.uint8)
bp = np.zeros(L, dtype=np.int32)
r = np.zeros(L); rpos = np.zeros(L + 1)
assert lib.rl_synth_panel(N, L, C.c_uint64(1), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                          bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                          rpos.ctypes.data_as(C.c_void_p)) == 0
budget = mem * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
wb = np.zeros(L + 2, dtype=np.int32)
W = lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499)
assert W > 0
sections = min(sections, W)
# posterior rows of the largest window (sum over targets of their derived sites in it), for the bounded windows
rows = max(int((seq[wb[w]:wb[w + 1]] == ord("1")).sum()) for w in range(min(W, sections)))
out = {"N": N, "L": L, "windows": int(W), "sections": sections, "ranks": ranks, "in_flight_per_rank": in_flight,
       "window_rows_kept": frac}
work = tempfile.mkdtemp()
try:
    d = os.path.join(work, "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0
    del seq
    hub = rdist.ThreadFabric.Hub(ranks)
    res, errs = [None] * ranks, [None] * ranks

    def body(rk):
        try:
            res[rk] = rdist.run_chunk_by_targets(d, 0, device=0, sections=list(range(sections)), in_flight=in_flight,
                                                 build_on_gpu=True,
                                                 window_rows=None if frac < 0 else (int(frac * rows / ranks) if frac < 1.0 else 0),  # (< 0: sized from the free HBM)
                                                 fabric=rdist.ThreadFabric(hub, rk, device=0))
        except BaseException as e:
            errs[rk] = e

    t0 = time.time()
    th = [threading.Thread(target=body, args=(rk,)) for rk in range(ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    out["wall_s"] = time.time() - t0
    assert errs == [None] * ranks, errs
    out["trees_kept"] = sum(sum(x.values()) for x in res)
    out["trees_per_s"] = out["trees_kept"] / out["wall_s"]
    out["md5"] = {f: hashlib.md5(open(os.path.join(d, "chunk_0", f), "rb").read()).hexdigest()
                  for f in ("out_0.anc", "out_0.mut")}
finally:
    shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out))
