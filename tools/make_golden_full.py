#!/usr/bin/env python3
"""Reference-held pins at FULL length for the configurations that had them at tile length only:

  c3_ends  BASELINE config #3 (N = 5000 x L = 500,000, seed 1, --memory 20): the two BOUNDARY windows, 0 and 266 --
           the reference's complete paint file of each (FastPainting::PaintSteppingStones, fast_painting.cpp:18-618,
           boundarySNP_begin = 0 / boundarySNP_end = L-1 special cases :60-69, :98-107, :150) and
           `Relate --mode BuildTopology` of each section (pipeline/BuildTopology.cpp:125-150)
           -> tests/golden/c3_ends.npz
  c2       config #2 (N = 1000 x L = 100,000, seed 1, --memory 5): the reference's `--mode Paint` of the whole chunk
           (md5 of every window's paint file) + BuildTopology of the first, a middle and the last section
           -> tests/golden/full_c2.npz
  c4       one chunk of config #4 as bench.py makes it (N = 2000 x L = 121,000, seed 1, --memory 1): the same
           -> tests/golden/full_c4.npz
  c5_first config #5 (N = 10,000 x L = 200,000, seed 1, --memory 25): the reference's complete paint file of window 0 and
           BuildTopology of section 0 (hours of the reference) -> tests/golden/c5_first.npz

Runs only in the build container (needs oracle/_ref, `make -C oracle ref`).  Resumable: every step leaves its output
in the work directory.   python tools/make_golden_full.py <c3_ends|c2|c4|c5_first> [workdir] [procs]
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil  # noqa: E402

CONFIGS = {
    "c3_ends": dict(N=5000, L=500000, mem=20.0, seed=1, work="/tmp/c3ref", out="c3_ends.npz", sections=None),
    "c2": dict(N=1000, L=100000, mem=5.0, seed=1, work="/tmp/c2ref", out="full_c2.npz", sections="first_mid_last"),
    "c4": dict(N=2000, L=121000, mem=1.0, seed=1, work="/tmp/c4ref", out="full_c4.npz", sections="first_mid_last"),
    "c5_first": dict(N=10000, L=200000, mem=25.0, seed=1, work="/tmp/c5ref", out="c5_first.npz", sections=None, records=True),
}


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return np.frombuffer(h.digest(), dtype=np.uint8)


def md5(b):
    return np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)


def chunk(work, N, L, mem, seed):
    """the chunk files, from this repo's generator (as tools/make_golden_c3.py and bench.py make them)"""
    from relate_amd import api
    lib = api.lib()
    d = os.path.join(work, "out")
    if os.path.exists(os.path.join(d, "parameters_c0.bin")):
        return
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = mem * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    W = lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499)
    os.makedirs(d, exist_ok=True)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0


def windows_of(work):
    p = np.fromfile(os.path.join(work, "out", "parameters_c0.bin"), dtype=np.int32)
    W = int(p[2]) - 1
    return W, p[3:3 + W + 1].copy()


def paint_windows(work, N, windows, procs):
    """the reference's relate_<w>.bin of the listed windows: `procs` harness processes, a range of targets each, every
    target painted over the whole chunk (ref_harness paint_windows); the parts concatenated ARE the files"""
    pdir = os.path.join(work, "out", "chunk_0", "paint")
    if all(os.path.exists(os.path.join(pdir, "relate_%d.bin" % w)) for w in windows):
        return
    cuts = [N * i // procs for i in range(procs + 1)]
    ps = []
    for i in range(procs):
        prefix = os.path.join(work, "ends_part%d" % i)
        if os.path.exists(prefix + ".done"):
            continue
        ps.append((prefix, subprocess.Popen(["nice", "-n", "10", rlutil.REF_HARNESS, "paint_windows", "out", "0",
                                             str(cuts[i]), str(cuts[i + 1]), prefix] + [str(w) for w in windows],
                                            cwd=work, stderr=open(prefix + ".log", "w"))))
    for prefix, p in ps:
        assert p.wait() == 0, prefix
        open(prefix + ".done", "w").close()
    os.makedirs(pdir, exist_ok=True)
    for w in windows:
        with open(os.path.join(pdir, "relate_%d.bin" % w), "wb") as fo:
            for i in range(procs):
                fo.write(open(os.path.join(work, "ends_part%d_%d.bin" % (i, w)), "rb").read())


def paint_all(work):
    """`Relate --mode Paint` of the unmodified reference: every window's paint file (pipeline/Paint.cpp:17-108)"""
    if os.path.exists(os.path.join(work, "paint.seconds")):
        return
    t0 = time.time()
    with open(os.path.join(work, "paint.log"), "w") as fh:
        subprocess.run(["nice", "-n", "10", rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", "0", "-o", "out"],
                       cwd=work, check=True, stdout=fh, stderr=subprocess.STDOUT)
    open(os.path.join(work, "paint.seconds"), "w").write("%.1f\n" % (time.time() - t0))


def build_topology(work, sections):
    ps = []
    for s in sections:
        if os.path.exists(os.path.join(work, "bt%d.seconds" % s)):
            continue
        fh = open(os.path.join(work, "bt%d.log" % s), "w")
        ps.append((s, time.time(), subprocess.Popen(
            ["nice", "-n", "10", rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section",
             str(s), "--last_section", str(s), "-o", "out"], cwd=work, stdout=fh, stderr=subprocess.STDOUT)))
    for s, t0, p in ps:
        assert p.wait() == 0, s
        open(os.path.join(work, "bt%d.seconds" % s), "w").write("%.1f\n" % (time.time() - t0))


def pack(cfg, work, sections, all_windows):
    d = os.path.join(work, "out")
    W, wb = windows_of(work)
    data = {"meta": np.array([cfg["N"], cfg["L"], W, cfg["seed"]], dtype=np.int64), "mem": np.array([cfg["mem"]]),
            "wb": wb, "sections": np.array(sections, dtype=np.int32)}
    for f in ["parameters_c0.bin", "chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.dist", "chunk_0.state"]:
        data["in_md5/" + f] = md5_file(os.path.join(d, f))
    pdir = os.path.join(d, "chunk_0", "paint")
    if all_windows:  # the whole Paint stage: every file
        data["paint_md5"] = np.load(os.path.join(work, "paint_all_md5.npy"))
        data["paint_size"] = np.load(os.path.join(work, "paint_all_size.npy"))
        data["reference_paint_s"] = np.array([float(open(os.path.join(work, "paint.seconds")).read())])
    for s in sections:
        pf = os.path.join(pdir, "relate_%d.bin" % s)
        if os.path.exists(pf):  # (the reference's BuildTopology removes the paint file of a section it has built)
            data["s%d/paint_md5" % s] = md5_file(pf)
            data["s%d/paint_size" % s] = np.array([os.path.getsize(pf)], dtype=np.int64)
        elif os.path.exists(os.path.join(work, "paint_md5_%d.npy" % s)):
            data["s%d/paint_md5" % s] = np.load(os.path.join(work, "paint_md5_%d.npy" % s))
            data["s%d/paint_size" % s] = np.load(os.path.join(work, "paint_size_%d.npy" % s))
        if cfg.get("records") and os.path.exists(pf):  # every target's record of the window, so that a test can check a subset
            import struct
            buf = open(pf, "rb").read()
            pos, lens, md5s = 0, [], []
            for _ in range(cfg["N"]):
                q = pos + 8
                for _stone in range(2):
                    one, n, bsnp, ls, K = struct.unpack_from("<QQifi", buf, q)
                    assert one == 1 and n == cfg["N"]
                    q += 28 + 8 * K
                lens.append(q - pos)
                md5s.append(md5(buf[pos:q]))
                pos = q
            assert pos == len(buf)
            data["s%d/record_len" % s] = np.array(lens, dtype=np.int32)
            data["s%d/record_md5" % s] = np.stack(md5s)
        anc = os.path.join(d, "chunk_0", "out_%d.anc" % s)
        mut = open(os.path.join(d, "chunk_0", "out_%d.mut" % s), "rb").read()
        data["s%d/anc_md5" % s] = md5_file(anc)
        data["s%d/anc_size" % s] = np.array([os.path.getsize(anc)], dtype=np.int64)
        data["s%d/mut_md5" % s] = md5(mut)
        data["s%d/mut" % s] = np.frombuffer(mut, dtype=np.uint8)
        _, trees = rlutil.parse_anc(anc)
        data["s%d/tree_pos" % s] = np.array([t[0] for t in trees], dtype=np.int32)
        data["s%d/tree_parent_md5" % s] = np.stack([md5(t[1].astype("<i4").tobytes()) for t in trees])
        data["s%d/reference_build_topology_s" % s] = np.array([float(open(os.path.join(work, "bt%d.seconds" % s)).read())])
    out = os.path.join(ROOT, "tests", "golden", cfg["out"])
    np.savez_compressed(out, **data)
    print("%s: W %d, sections %s, %.1f KB" % (cfg["out"], W, sections, os.path.getsize(out) / 1e3))


def keep_paint_md5(work, sections):
    """(the paint files' md5 before BuildTopology, which deletes the file of a section it has built)"""
    pdir = os.path.join(work, "out", "chunk_0", "paint")
    for s in sections:
        pf = os.path.join(pdir, "relate_%d.bin" % s)
        if os.path.exists(pf) and not os.path.exists(os.path.join(work, "paint_md5_%d.npy" % s)):
            np.save(os.path.join(work, "paint_md5_%d.npy" % s), md5_file(pf))
            np.save(os.path.join(work, "paint_size_%d.npy" % s), np.array([os.path.getsize(pf)], dtype=np.int64))


if __name__ == "__main__":
    assert rlutil.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    name = sys.argv[1]
    cfg = CONFIGS[name]
    work = sys.argv[2] if len(sys.argv) > 2 else cfg["work"]
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    os.makedirs(work, exist_ok=True)
    t0 = time.time()
    chunk(work, cfg["N"], cfg["L"], cfg["mem"], cfg["seed"])
    W, wb = windows_of(work)
    print("[%s] chunk: W = %d, %.0f s" % (name, W, time.time() - t0), flush=True)
    if name in ("c3_ends", "c5_first"):
        sections = [0, W - 1] if name == "c3_ends" else [0]
        paint_windows(work, cfg["N"], sections, procs)
        print("[%s] paint files of windows %s: %.0f s" % (name, sections, time.time() - t0), flush=True)
        all_windows = False
    else:
        sections = sorted(set([0, W // 2, W - 1]))
        paint_all(work)
        print("[%s] reference Paint: %.0f s" % (name, time.time() - t0), flush=True)
        all_windows = True
        if not os.path.exists(os.path.join(work, "paint_all_md5.npy")):
            pdir = os.path.join(work, "out", "chunk_0", "paint")
            np.save(os.path.join(work, "paint_all_md5.npy"), np.stack([md5_file(os.path.join(pdir, "relate_%d.bin" % w)) for w in range(W)]))
            np.save(os.path.join(work, "paint_all_size.npy"),
                    np.array([os.path.getsize(os.path.join(pdir, "relate_%d.bin" % w)) for w in range(W)], dtype=np.int64))
    keep_paint_md5(work, sections)
    build_topology(work, sections)
    print("[%s] BuildTopology of sections %s: %.0f s" % (name, sections, time.time() - t0), flush=True)
    pack(cfg, work, sections, all_windows)
