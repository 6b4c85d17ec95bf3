#!/usr/bin/env python3
"""BASELINE.json config #4 as a JOB on one GPU: synthetic N = 2000 haplotypes x 5M SNPs as a
.haps / .sample / map text data set -> `Relate --mode MakeChunks --memory 1` (~50 chunks) -> every chunk through
Paint + BuildTopology (fused stage, all sections) + FindEquivalentBranches, the route of
scripts/RelateParallel/RelateParallel.sh:216-262 with one rank (relate_amd.dist.run_chunks) -- the N = 1 anchor of the
1 / 2 / 4 / 8-GPU curve -- and from the per-chunk seconds the time the same deal takes on G ranks (chunks are
independent: no collective, no barrier; MakeChunks is the serial part).

    python tools/c4_job_one_gpu.py [N=2000] [L=5000000] [memory=1] [workdir] [max_chunks]

Prints one JSON line.  ~25 GB of text + ~45 GB of chunk files in workdir (removed at the end)."""
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relate_amd import api  # noqa: E402
from relate_amd import dist as rdist  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
MEM = sys.argv[3] if len(sys.argv) > 3 else "1"
work = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else tempfile.mkdtemp(prefix="c4job_")
max_chunks = int(sys.argv[5]) if len(sys.argv) > 5 else 0
os.makedirs(work, exist_ok=True)
out = {"N": N, "L": L, "memory": float(MEM), "workdir_free_GB_before": shutil.disk_usage(work).free / 1e9}
lib = api.lib()
CLI = os.path.join(ROOT, "relate_amd", "Relate")


def write_dataset():
    """the bench's block-coalescent generator, a piece of PIECE SNPs at a time (seed 1 + piece), as text: one line per
    SNP `1 rs<s> <bp> A T a_0 ... a_{N-1}`; positions 100 bp apart, a uniform 1 cM/Mb map"""
    PIECE = 250000
    t0 = time.time()
    lut = np.zeros((2, 2), dtype=np.uint8)
    lut[0] = (ord("0"), ord(" "))
    lut[1] = (ord("1"), ord(" "))
    nbytes = 0
    bp_base = 0
    with open(os.path.join(work, "c4.haps"), "wb", buffering=1 << 24) as f:
        for piece, s0 in enumerate(range(0, L, PIECE)):
            n = min(PIECE, L - s0)
            seq = np.zeros((n, N), dtype=np.uint8)
            bp = np.zeros(n, dtype=np.int32)
            r = np.zeros(n)
            rpos = np.zeros(n + 1)
            assert lib.rl_synth_panel(N, n, C.c_uint64(1 + piece), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                                      bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                      rpos.ctypes.data_as(C.c_void_p)) == 0
            body = lut[seq - ord("0")].reshape(n, 2 * N)  # "a_0 a_1 ... a_{N-1} "
            body[:, -1] = ord("\n")
            pos = bp.astype(np.int64) + bp_base
            # (lines of equal header length go out as one 2-D block: no per-line copies)
            heads = [b"1 rs%d %d A T " % (s0 + i, pos[i]) for i in range(n)]
            lens = np.fromiter((len(h) for h in heads), dtype=np.int64, count=n)
            cuts = [0] + [int(x) + 1 for x in np.nonzero(np.diff(lens))[0]] + [n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                for lo in range(a, b, 50000):
                    hi = min(b, lo + 50000)
                    h = int(lens[lo])
                    arr = np.empty((hi - lo, h + 2 * N), dtype=np.uint8)
                    arr[:, :h] = np.frombuffer(b"".join(heads[lo:hi]), dtype=np.uint8).reshape(hi - lo, h)
                    arr[:, h:] = body[lo:hi]
                    f.write(arr.data)
                    nbytes += arr.size
            bp_base = int(pos[-1]) + 100
    with open(os.path.join(work, "c4.sample"), "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(N // 2):
            f.write("id%d id%d 0\n" % (i, i))
    with open(os.path.join(work, "c4.map"), "w") as f:
        f.write("pos COMBINED_rate Genetic_Map\n")
        for bp in range(0, bp_base + 100000, 50000):
            f.write("%d 1.0 %.6f\n" % (bp, bp * 1e-6))
    out["dataset"] = {"haps_GB": nbytes / 1e9, "write_s": time.time() - t0, "last_bp": bp_base}


try:
    write_dataset()
    t0 = time.time()
    p = subprocess.run([CLI, "--mode", "MakeChunks", "--haps", "c4.haps", "--sample", "c4.sample", "--map", "c4.map",
                        "--memory", MEM, "-o", "job"], cwd=work, stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_TIMING="1", RELATE_AMD_CHUNK_BITS="1"))
    assert p.returncode == 0, p.stderr.decode()[-800:]
    out["makechunks_s"] = time.time() - t0
    out["makechunks_lines"] = [l.strip() for l in p.stderr.decode().split("\n") if l.startswith("[make")][:8]
    os.remove(os.path.join(work, "c4.haps"))
    job = os.path.join(work, "job")
    par = rdist.read_parameters(job)
    chunks = par["num_chunks"]
    out["chunks"] = chunks
    out["job_files_GB"] = sum(os.path.getsize(os.path.join(job, f)) for f in os.listdir(job)) / 1e9
    todo = list(range(chunks if not max_chunks else min(chunks, max_chunks)))
    timings = []
    sections = trees = 0
    anc_bytes = 0
    t0 = time.time()
    for c in todo:  # one rank: every chunk, start to end (run_chunks' loop; a chunk's files are counted and removed
        rdist.run_chunks(job, chunks=[c], timings=timings)  # before the next one: ~1.7 GB of .anc per chunk)
        d = os.path.join(job, "chunk_%d" % c)
        for fn in os.listdir(d):
            if fn.endswith(".anc"):
                sections += 1
                anc_bytes += os.path.getsize(os.path.join(d, fn))
                with open(os.path.join(d, fn), "rb") as fh:
                    trees += int(np.frombuffer(fh.read(9)[5:9], dtype=np.uint32)[0])
        shutil.rmtree(d, ignore_errors=True)
        for fn in os.listdir(job):
            if fn.startswith("chunk_%d." % c):
                os.remove(os.path.join(job, fn))
    out["chunks_run"] = len(todo)
    out["run_chunks_s"] = time.time() - t0
    per = {}
    for c, stage, secs in timings:
        per.setdefault(c, {})[stage] = secs
    out["per_chunk_s"] = {str(c): {k: round(v, 2) for k, v in st.items()} for c, st in sorted(per.items())}
    tot = {}
    for st in per.values():
        for k, v in st.items():
            tot[k] = tot.get(k, 0.0) + v
    out["stage_totals_s"] = {k: round(v, 1) for k, v in tot.items()}
    out["sections"] = sections
    out["trees_kept"] = trees
    out["anc_GB"] = anc_bytes / 1e9
    chunk_s = [sum(per[c].values()) for c in todo]
    scale = chunks / float(len(todo))
    out["job_one_gpu_s"] = out["makechunks_s"] + out["run_chunks_s"] * scale
    # the same deal (chunk c on rank c mod G, relate_amd.dist.shard) with this run's per-chunk seconds: MakeChunks is
    # serial and host-only, the ranks share nothing but the input directory
    proj = {}
    for G in (1, 2, 4, 8):
        per_rank = [sum(chunk_s[i] for i in range(len(chunk_s)) if i % G == r) for r in range(G)]
        proj[str(G)] = {"chunks_part_s": round(max(per_rank) * (scale if G == 1 else 1.0), 1),
                        "with_makechunks_s": round(out["makechunks_s"] + max(per_rank) * (scale if G == 1 else 1.0), 1)}
    out["projected_ranks"] = proj
    out["serial_host_parts"] = "MakeChunks (text parse + chunk files, one process before any rank starts); per chunk on its " \
                               "rank: reading the chunk files, FindEquivalentBranches (host threads)"
finally:
    if len(sys.argv) <= 4 or sys.argv[4] == "-":
        shutil.rmtree(work, ignore_errors=True)
    else:
        shutil.rmtree(os.path.join(work, "job"), ignore_errors=True)
        for fn in ("c4.haps", "c4.sample", "c4.map"):
            try:
                os.remove(os.path.join(work, fn))
            except OSError:
                pass
print(json.dumps(out))
