#!/usr/bin/env python3
"""FindEquivalentBranches of this library against the reference binary on a synthetic chunk larger than the
committed fixtures (needs oracle/_ref/Relate, i.e. a container with /root/reference; CPU only):

    python tools/check_feb_against_reference.py [N L memory_GB seed]

The reference paints and builds the trees, then both run FindEquivalentBranches on copies of its .anc files."""
import ctypes as C, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
import rlutil

N, L, mem, seed = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 3000, 0.01, 5)
lib = api.lib()
seq = np.zeros((L, N), dtype=np.uint8); bp = np.zeros(L, dtype=np.int32); r = np.zeros(L); rpos = np.zeros(L + 1)
assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                          bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p)) == 0
budget = mem * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
wb = np.zeros(L + 2, dtype=np.int32)
W = lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499)
assert W > 1, W
work = tempfile.mkdtemp()
try:
    d = os.path.join(work, "out"); os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0
    run = lambda exe, *a: subprocess.run([exe] + list(a), cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    run(rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", "0", "-o", "out")
    run(rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section", str(W - 1), "-o", "out")
    shutil.copytree(os.path.join(d, "chunk_0"), os.path.join(work, "before"))
    t0 = time.time()
    run(rlutil.REF_RELATE, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out")
    t_ref = time.time() - t0
    ref = {f: open(os.path.join(d, "chunk_0", f), "rb").read() for f in os.listdir(os.path.join(d, "chunk_0")) if f.endswith(".anc")}
    for f in ref:
        shutil.copy(os.path.join(work, "before", f), os.path.join(d, "chunk_0", f))
    t0 = time.time()
    run(os.path.join(ROOT, "relate_amd", "Relate"), "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out")
    t_own = time.time() - t0
    bad = [f for f in ref if open(os.path.join(d, "chunk_0", f), "rb").read() != ref[f]]
    changed = sum(open(os.path.join(work, "before", f), "rb").read() != ref[f] for f in ref)
    print("N=%d L=%d windows=%d: %d .anc files, %d rewritten by the stage, %d differ from the reference's; "
          "the stage took the reference %.2f s, this library %.2f s" % (N, L, W, len(ref), changed, len(bad), t_ref, t_own))
    sys.exit(1 if bad else 0)
finally:
    shutil.rmtree(work, ignore_errors=True)
