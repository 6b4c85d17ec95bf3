#!/usr/bin/env python3
"""The C3 chunk (synthetic N = 5000 x L = 500000, 267 sections) through `Relate --mode PaintBuildTopology` (stepping
stones kept in HBM, no paint files): Paint of the whole chunk + BuildTopology of sections [0, sections).

    python tools/chunk_c3_fused.py sections [N L memory_GB]

Prints one JSON line: wall-clock, trees, trees/s, the stage's own phase lines."""
import ctypes as C, json, os, re, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from relate_amd import api
import rlutil

sections = int(sys.argv[1])
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
sections = min(sections, W)
out = {"N": N, "L": L, "windows": int(W), "sections": sections}
work = tempfile.mkdtemp()
try:
    d = os.path.join(work, "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0
    del seq
    exe = os.environ.get("RELATE_EXE") or os.path.join(ROOT, "relate_amd", "Relate")  # (RELATE_EXE: another build, A/B runs)
    t0 = time.time()
    # C3_FUSED_FEB=1: FindEquivalentBranches fused behind the stage (every .anc written once, as that stage leaves it)
    fused_feb = ["--find_equivalent_branches"] if os.environ.get("C3_FUSED_FEB") and sections == W else []
    out["fused_find_equivalent_branches"] = bool(fused_feb)
    p = subprocess.run([exe, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--first_section", "0",
                        "--last_section", str(sections - 1), "-o", "out"] + fused_feb, cwd=work, stderr=subprocess.PIPE,
                       env=dict(os.environ) if os.environ.get("C3_NO_TIMING") else dict(os.environ, RELATE_AMD_TIMING="1"))
    out["wall_s"] = time.time() - t0
    import resource
    ru = resource.getrusage(resource.RUSAGE_CHILDREN)
    out["stage_cpu_s"] = {"user": round(ru.ru_utime, 1), "sys": round(ru.ru_stime, 1),
                          "cores_busy_on_average": round((ru.ru_utime + ru.ru_stime) / out["wall_s"], 1)}
    try:
        out["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception:
        pass
    err = p.stderr.decode()
    if os.environ.get("C3_KEEP_STDERR"):  # (everything but the per-tree and per-window lines)
        with open(os.path.join(ROOT, os.environ["C3_KEEP_STDERR"]), "w") as fh:
            fh.write("\n".join(l for l in err.replace("\r", "\n").split("\n")
                               if l.strip() and "[gpu tree builder]" not in l and not re.match(r"^\[\d+/\d+\]$", l.strip())))
    assert p.returncode == 0, err[-600:]
    trees = 0
    for l in err.split("\n"):
        m = re.search(r"(\d+) trees kept of (\d+) built", l)
        if m:
            trees += int(m.group(2))
    # trees kept: the count in every .anc header (byte has_ages, u32 N, u32 trees)
    import struct
    kept = 0
    for f in os.listdir(os.path.join(d, "chunk_0")):
        if f.endswith(".anc"):
            with open(os.path.join(d, "chunk_0", f), "rb") as fh:
                kept += struct.unpack("<I", fh.read(9)[5:9])[0]
    out["trees_kept"] = kept
    # md5 of a few sections' files: held against tools/verify_c3_sections.py's (the same sections rebuilt one per call,
    # whole window resident, host tree builder), committed as profiles/r04_c3_section_md5.json -- bench.py compares
    import hashlib
    out["section_md5"] = {}
    for sct in (0, W // 2, W - 1):
        for e in ("anc", "mut"):
            fn = os.path.join(d, "chunk_0", "out_%d.%s" % (sct, e))
            if os.path.exists(fn):
                out["section_md5"]["out_%d.%s" % (sct, e)] = hashlib.md5(open(fn, "rb").read()).hexdigest()
    # (built = kept + the rebuilt trees that were dropped again, anc_builder.cpp:621-630: counted by the stage's
    #  per-section lines, i.e. only in a run with the timing lines; never one under the other's name)
    out["trees_built"] = trees if trees else None
    out["trees_per_s"] = (trees if trees else kept) / out["wall_s"]
    out["trees_per_s_counts"] = "built" if trees else "kept"
    out["anc_GB"] = sum(os.path.getsize(os.path.join(d, "chunk_0", f)) for f in os.listdir(os.path.join(d, "chunk_0"))
                        if f.endswith(".anc")) / 1e9
    out["fused_feb_lines"] = [l.strip() for l in err.split("\n") if "find equivalent branches, fused" in l]
    out["stage_lines"] = [l.strip() for l in err.split("\n") if l.startswith("[") and "find equivalent" not in l and "\r" not in l and "[window " not in l and "tree sequence" not in l
                          and "[tree builder workers]" not in l and "[gpu tree builder]" not in l][:12]
    out["stage_summary"] = [l.strip() for l in err.split("\n") if l.startswith("[stage] sections") or l.startswith("[stage] context")]
    out["builder_worker_launches"] = [l.strip() for l in err.split("\n") if "[tree builder workers]" in l][:24] + [l.strip() for l in err.split("\n") if "waiting for RePaint:" in l][:60]
    out["builder_host_side"] = [l.strip() for l in err.split("\n") if "host ms per tree" in l][:6]
    acc, ntr = {}, 0
    for l in err.split("\n"):
        if "[gpu tree builder]" in l and "us:" in l:
            ntr += 1
            for m in re.finditer(r"([a-z_+ ]+?) (\d+)(?= |$)", l.split("us:")[1]):
                acc[m.group(1).strip()] = acc.get(m.group(1).strip(), 0) + int(m.group(2))
    if ntr:
        out["gpu_builder_ms_per_tree"] = {k: round(v / ntr / 1000.0, 2) for k, v in acc.items()}
    wl = [l.strip() for l in err.split("\n") if l.startswith("[window ")]
    out["window_lines"] = wl[:3] + wl[len(wl) // 2: len(wl) // 2 + 3]
    # over ALL windows: what a section waited for RePaint's turn, its launches, its matrices (means, seconds per window)
    agg = {"turn": [], "launches": [], "matrices": [], "rows": []}
    for l in wl:
        m = re.search(r"rows \+ uploads ([\d.]+), waiting for RePaint's turn ([\d.]+), RePaint launches ([\d.]+), matrices ([\d.]+)", l)
        if m:
            agg["rows"].append(float(m.group(1))); agg["turn"].append(float(m.group(2)))
            agg["launches"].append(float(m.group(3))); agg["matrices"].append(float(m.group(4)))
    if agg["turn"]:
        out["per_window_mean_s"] = {k: round(sum(v) / len(v), 2) for k, v in agg.items()}
        out["per_window_max_wait_s"] = max(agg["turn"])
    ts = [l for l in err.split("\n") if "[tree sequence]" in l]
    tb = [float(x) for l in ts for x in re.findall(r"tree builds ([\d.]+) s", l)]
    tm = [float(x) for l in ts for x in re.findall(r"distance matrices ([\d.]+) s", l)]
    if tb:
        out["per_section_mean_s"] = {"tree_builds": round(sum(tb) / len(tb), 1), "distance_matrices_incl_repaint": round(sum(tm) / len(tm), 1)}
    out["one_section"] = [l.strip() for l in err.split("\n") if "[tree sequence]" in l][:1]
    # when each section's window was open and when its trees were done, seconds after the stage began
    sec = [(int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))) for m in
           re.finditer(r"\[section (\d+)\] turn (\d+), \d+ SNPs: window open ([\d.]+) s after the stage began, trees built at ([\d.]+) s", err)]
    if sec:
        ends = sorted(x[3] for x in sec)
        out["sections_timeline"] = {"first_window_open_s": min(x[2] for x in sec), "last_first_round_open_s": max(x[2] for x in sec if x[1] < 134),
                                    "sections_done_by_s": {str(q): ends[min(len(ends) - 1, int(q * len(ends) / 100))] for q in (10, 25, 50, 75, 90, 95, 99)},
                                    "last_done_s": ends[-1],
                                    "open_sections_at_s": {str(t): sum(1 for x in sec if x[2] <= t < x[3]) for t in range(0, int(ends[-1]) + 10, 10)}}
    if os.environ.get("C3_FEB") and sections == W:
        # the stage downstream on the same files (pipeline/FindEquivalentBranches.cpp:78-145): reads every .anc of the
        # chunk, associates the branches of neighbouring trees, rewrites the files in place
        t0 = time.time()
        p = subprocess.run([exe, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out"], cwd=work,
                           stderr=subprocess.PIPE, env=dict(os.environ, RELATE_AMD_TIMING="1"))
        out["find_equivalent_branches_s"] = time.time() - t0
        out["find_equivalent_branches_rc"] = p.returncode
        out["find_equivalent_branches_lines"] = [l.strip() for l in p.stderr.decode().split("\n") if l.strip()][-8:]
        out["feb_md5"] = {}
        for sct in (0, W // 2, W - 1):
            fn = os.path.join(d, "chunk_0", "out_%d.anc" % sct)
            if os.path.exists(fn):
                out["feb_md5"]["out_%d.anc" % sct] = hashlib.md5(open(fn, "rb").read()).hexdigest()
finally:
    shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out))
