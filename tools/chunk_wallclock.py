#!/usr/bin/env python3
"""Chunk wall-clock of the two stages through the drop-in CLI (files in -> files out).

    python tools/chunk_wallclock.py N L memory_GB [ref]

Writes a synthetic chunk (chunk_0.*, parameters_c0.bin), runs
`relate_amd/Relate --mode Paint` and `--mode BuildTopology` over all sections and
prints one JSON line with the wall-clock of each stage and per-kernel times of
one window (RePaint K2, one distance matrix K3).  With a 4th argument `ref` the
unmodified reference binary (oracle/_ref/Relate, build container only) is timed
on the same chunk instead; `md5_of_all_outputs` must agree between the two."""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil

N, L, mem = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
use_ref = len(sys.argv) > 4 and sys.argv[4] == "ref"
budget = mem * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
ch = rlutil.synth_chunk(N, L, seed=1, budget=budget)
out = {"N": N, "L": L, "windows": int(ch.W)}
with tempfile.TemporaryDirectory() as work:
    ch.write(os.path.join(work, "out"))
    exe = rlutil.REF_RELATE if use_ref else os.path.join(ROOT, "relate_amd", "Relate")
    tag = "reference_cpu_1thread" if use_ref else "mi355x"
    t0 = time.time()
    subprocess.run([exe, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=work, check=True, stderr=subprocess.PIPE)
    t1 = time.time()
    subprocess.run([exe, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                    str(ch.W - 1), "-o", "out"], cwd=work, check=True, stderr=subprocess.PIPE)
    t2 = time.time()
    out[tag] = {"paint_s": t1 - t0, "build_topology_s": t2 - t1}
    h = hashlib.md5()
    ntrees = 0
    for w in range(ch.W):
        for fn in ("paint/relate_%d.bin" % w, "out_%d.anc" % w, "out_%d.mut" % w):
            h.update(open(os.path.join(work, "out", "chunk_0", fn), "rb").read())
        ntrees += len(rlutil.parse_anc(os.path.join(work, "out", "chunk_0", "out_%d.anc" % w))[1])
    out["md5_of_all_outputs"] = h.hexdigest()
    out["trees"] = ntrees
    if not use_ref:
        from relate_amd import api
        ctx = api.Context(); ctx.load_chunk(os.path.join(work, "out"), 0)
        w = ch.W // 2
        win = ctx.open_window(w, os.path.join(work, "out", "chunk_0", "paint", "relate_%d.bin" % w), int(ch.wb[w]))
        win.matrix(int(ch.wb[w]))
        rows = sum(win.rows(n) for n in range(N))
        out["k2_repaint_ms_one_window"] = win.repaint_ms
        out["k2_rows"] = rows
        out["k2_topology_GBps"] = rows * N * 4 / (win.repaint_ms * 1e-3) / 1e9
        out["k3_matrix_ms"] = win.matrix_ms
        out["k3_GBps"] = 12.0 * N * N / (win.matrix_ms * 1e-3) / 1e9
print(json.dumps(out))
