#!/usr/bin/env python3
"""Chunk wall-clock sample at N = 5000 through the drop-in CLI (files in -> files out).

    python tools/chunk_wallclock_big.py N L memory_GB sections

Synthetic chunk (the bench's generator) -> chunk files -> `Relate --mode Paint` (whole chunk) and
`Relate --mode BuildTopology` for the first `sections` windows; prints one JSON line.  BuildTopology's
host tree building (MinMatch, O(N^2) per tree) dominates at this N, so only a few sections are timed."""
import ctypes as C, hashlib, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
import rlutil

N, L, mem, sections = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
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
out = {"N": N, "L": L, "windows": int(W), "sections_timed": min(sections, W)}
with tempfile.TemporaryDirectory() as work:
    d = os.path.join(work, "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0
    del seq
    exe = os.environ.get("RELATE_EXE") or os.path.join(ROOT, "relate_amd", "Relate")  # (RELATE_EXE: another build, A/B runs)
    if len(sys.argv) > 5 and sys.argv[5] == "ref":  # the unmodified reference binary on the same chunk (container only)
        exe = rlutil.REF_RELATE
    t0 = time.time()
    p = subprocess.run([exe, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=work, stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()[-400:]
    out["paint_stage_phases"] = [l.strip() for l in p.stderr.decode().split("\n") if l.startswith("[paint stage]")]
    t1 = time.time()
    out["paint_stage_s"] = t1 - t0
    out["paint_files_GB"] = sum(os.path.getsize(os.path.join(d, "chunk_0", "paint", f))
                                for f in os.listdir(os.path.join(d, "chunk_0", "paint"))) / 1e9
    t1 = time.time()
    if sections > 0:
        pre = []
        if os.environ.get("CHUNK_PMC"):  # counters of the BuildTopology process: CHUNK_PMC="out_dir:COUNTER COUNTER ..."
            odir, ctrs = os.environ["CHUNK_PMC"].split(":")
            pre = ["rocprofv3", "--pmc"] + ctrs.split() + ["--kernel-trace", "-d", os.path.abspath(odir), "-o", "bt", "--"]
        elif os.environ.get("CHUNK_ROCPROF"):  # kernel trace of the BuildTopology process (the program itself after --)
            pre = ["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.abspath(os.environ["CHUNK_ROCPROF"]), "-o", "bt",
                   "--"]
        p = subprocess.run(pre + [exe, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                            str(min(sections, W) - 1), "-o", "out"], cwd=work, stderr=subprocess.PIPE,
                           env=dict(os.environ, RELATE_AMD_TIMING="1"))
        assert p.returncode == 0, p.stderr.decode()[-400:]
        out["build_topology_phases"] = [l.strip() for l in p.stderr.decode().split("\n") if "[tree sequence]" in l]
        out["builder_host_side"] = [l.strip() for l in p.stderr.decode().split("\n") if "host ms per tree" in l][:4]
        launches = [l for l in p.stderr.decode().split("\n") if "[tree builder launch]" in l]
        if launches:
            import re as _re
            sizes = [int(_re.search(r"launch\] (\d+) trees", l).group(1)) for l in launches]
            out["builder_launches"] = {"launches": len(sizes), "mean_trees_per_launch": sum(sizes) / len(sizes)}
        import re
        acc, ntr = {}, 0
        for l in p.stderr.decode().split("\n"):
            if "[gpu tree builder]" in l and "us:" in l:
                ntr += 1
                for m in re.finditer(r"([a-z_+ ]+?) (\d+)(?= |$)", l.split("us:")[1]):
                    acc[m.group(1).strip()] = acc.get(m.group(1).strip(), 0) + int(m.group(2))
        if ntr:
            out["gpu_builder_ms_per_tree"] = {k: round(v / ntr / 1000.0, 2) for k, v in acc.items()}
            out["gpu_builder_trees_timed"] = ntr
    t2 = time.time()
    out["build_topology_s"] = t2 - t1
    trees = snps = 0
    for w in range(min(sections, W)):
        trees += len(rlutil.parse_anc(os.path.join(d, "chunk_0", "out_%d.anc" % w))[1])
        snps += int(wb[w + 1] - wb[w])
    out["trees"] = trees
    md5 = lambda fn: hashlib.md5(open(os.path.join(d, "chunk_0", fn), "rb").read()).hexdigest()
    out["md5"] = {"paint/relate_0.bin": md5("paint/relate_0.bin")}
    for w in range(min(sections, W)):
        out["md5"]["out_%d.anc" % w] = md5("out_%d.anc" % w)
        out["md5"]["out_%d.mut" % w] = md5("out_%d.mut" % w)
    out["snps_in_timed_sections"] = snps
    out["host_threads"] = os.cpu_count()
print(json.dumps(out))
