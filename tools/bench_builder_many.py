#!/usr/bin/env python3
"""Trees per second of the resident tree-builder workers when MANY builders build side by side (what
do two workers on one CU give?).  The matrices of one real build (RELATE_AMD_TEST_MM_DUMP, tools/bench_builder_variants.sh
writes them: <dir>/d_<k>.bin, cf_<k>.bin) are built by `builders` builders x `reps` repetitions with `workers`
resident workgroups; every build must come out the same tree.

    python tools/bench_builder_many.py <dumpdir> <k> builders:workers[:reps] [builders:workers[:reps] ...]

One JSON line per configuration.  RELATE_AMD_BUILD_OCC=1 keeps one workgroup per CU (256 registers per lane)."""
import ctypes as C, hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relate_amd import api
d, k = sys.argv[1], int(sys.argv[2])
dm = np.fromfile(os.path.join(d, "d_%d.bin" % k), np.float32)
N = int(round(dm.size ** 0.5))
cf = os.path.join(d, "cf_%d.bin" % k)
pr = np.fromfile(cf, np.float32) if os.path.exists(cf) else None
lib = api.lib()
lib.rl_debug_builder_throughput.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
for cfg in sys.argv[3:]:
    parts = [int(x) for x in cfg.split(":")]
    builders, workers, reps = parts[0], parts[1], (parts[2] if len(parts) > 2 else 4)
    secs, bad = C.c_double(0), C.c_int(0)
    first = np.zeros((reps, 2 * N - 1), np.int32)
    t0 = time.time()
    rc = lib.rl_debug_builder_throughput(N, 0.001, 0, builders, reps, workers, dm.ctypes.data_as(C.c_void_p),
                                         pr.ctypes.data_as(C.c_void_p) if pr is not None else None, C.byref(secs),
                                         C.byref(bad), first.ctypes.data_as(C.c_void_p))
    print(json.dumps({"N": N, "builders": builders, "workers": workers, "reps": reps, "rc": rc,
                      "occ_env": os.environ.get("RELATE_AMD_BUILD_OCC"), "seconds": round(secs.value, 3),
                      "trees_per_s": round(builders * reps / max(secs.value, 1e-9), 1), "mismatches": bad.value,
                      "ms_per_tree_per_worker": round(1e3 * secs.value * min(workers or builders, builders) / (builders * reps), 1),
                      "first_tree_md5": hashlib.md5(first[0].tobytes()).hexdigest(), "total_s": round(time.time() - t0, 1),
                      "error": lib.rl_last_error().decode() if rc else None}), flush=True)
