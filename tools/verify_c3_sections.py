#!/usr/bin/env python3
"""Is the full-size C3 run right?  The stage that builds the whole chunk (tools/chunk_c3_fused.py: 134 sections in
flight, every window a 1/20 - 1/31 part resident and re-painted ~37 times, the trees on the device workers) and a
DIFFERENT schedule of the same path -- one section at a time, its whole window resident (one RePaint launch), the
trees by the host's MinMatch -- must write the same bytes: md5 of out_<s>.anc / .mut for a few sections, both ways.

    python tools/verify_c3_sections.py [sections, default 0,133,266] [N L memory_GB]

Prints one JSON line {"sections": [...], "stage_md5": {...}, "check_md5": {...}, "verified_sections": [...]}.
(The unmodified reference binary on the same chunk: Paint ~2.5 h + ~15 min per section on one core; its section-0
files for the L = 20000 cut of this chunk are the md5s in tools/chunk_wallclock_big.py / BENCH_r03.)"""
import ctypes as C, hashlib, json, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api

want = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 133, 266]
N, L, mem = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (5000, 500000, 20.0)
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
want = sorted(set(min(s, W - 1) for s in want))
out = {"N": N, "L": L, "windows": int(W), "sections": want}
work = tempfile.mkdtemp()
exe = os.path.join(ROOT, "relate_amd", "Relate")


def md5s(d, s):
    return {"out_%d.%s" % (s, e): hashlib.md5(open(os.path.join(d, "chunk_0", "out_%d.%s" % (s, e)), "rb").read()).hexdigest()
            for e in ("anc", "mut")}


try:
    d = os.path.join(work, "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0
    del seq
    # 1. the stage as bench.py runs it: all sections in one call
    t0 = time.time()
    p = subprocess.run([exe, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", str(W - 1), "-o", "out"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-600:]
    out["stage_wall_s"] = time.time() - t0
    out["stage_md5"] = {}
    for s in want:
        out["stage_md5"].update(md5s(d, s))
        for e in ("anc", "mut"):
            os.rename(os.path.join(d, "chunk_0", "out_%d.%s" % (s, e)), os.path.join(work, "stage_%d.%s" % (s, e)))
    shutil.rmtree(os.path.join(d, "chunk_0"))
    # 2. another schedule: one section per call, the whole window resident, the host's tree builder
    out["check_md5"] = {}
    t0 = time.time()
    for s in want:
        p = subprocess.run([exe, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--first_section", str(s),
                            "--last_section", str(s), "-o", "out"], cwd=work, stderr=subprocess.PIPE,
                           env=dict(os.environ, RELATE_AMD_GPU_BUILD="0", RELATE_AMD_WINDOW_ROWS="0"))
        assert p.returncode == 0, p.stderr.decode()[-600:]
        out["check_md5"].update(md5s(d, s))
    out["check_wall_s"] = time.time() - t0
    out["verified_sections"] = [s for s in want if all(out["stage_md5"]["out_%d.%s" % (s, e)] == out["check_md5"]["out_%d.%s" % (s, e)]
                                                       for e in ("anc", "mut"))]
    out["against"] = "the same sections rebuilt one per call with the whole window resident (one RePaint launch) and the host's MinMatch"
finally:
    shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out))
sys.exit(0 if out.get("verified_sections") == want else 1)
