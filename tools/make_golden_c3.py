#!/usr/bin/env python3
"""tests/golden/c3_full.npz: the HEADLINE configuration (BASELINE.json config #3: synthetic N = 5000 x L = 500,000,
seed 1, --memory 20, 267 windows) against the unmodified reference at FULL length.

Runs only in the build container (needs /root/reference compiled into oracle/_ref by `make -C oracle ref`); hours
of CPU.  Resumable: every step leaves its output in the work directory and is skipped when it is there.

  step 1  chunk files of the seed-1 C3 chunk (this repo's generator; md5s kept)
  step 2  ref_harness paint_targets: FastPainting::PaintSteppingStones at full length for 16 targets
          (fast_painting.cpp:18-618) -> md5 of every (window, target) record, both logscales of every record in full
  step 3  ref_harness paint_window 133: the reference's complete paint file of ONE window (all 5000 targets; 7
          processes x ~30 min) -> md5, size, head
  step 4  Relate --mode BuildTopology --first_section 133 --last_section 133 on that file
          (pipeline/BuildTopology.cpp:125-150) -> md5 of .anc / .mut, the .mut in full, md5 of every parent array,
          three arrays in full
  step 5  ref_harness matrix 133 at three SNPs -> md5 of each 5000 x 5000 matrix, 8 rows in full;
          ref_harness repaint_targets 133 for 4 targets -> md5 of their posterior rows, logscales in full

    python tools/make_golden_c3.py [workdir=/tmp/c3ref] [step ...]
"""
import hashlib
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil  # noqa: E402

N, L, MEM, SEED, W_PIN = 5000, 500000, 20.0, 1, 133
TARGETS = [0, 1, 63, 64, 624, 1250, 1999, 2047, 2500, 3333, 4095, 4096, 4500, 4937, 4998, 4999]
REPAINT_TARGETS = [0, 2047, 3333, 4999]
MATRIX_ROWS = (0, 1, 2, 1250, 2500, 3750, 4998, 4999)
md5 = lambda b: np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)


def chunk(work):
    import ctypes as C
    from relate_amd import api
    lib = api.lib()
    d = os.path.join(work, "out")
    if os.path.exists(os.path.join(d, "parameters_c0.bin")):
        return
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(SEED), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = MEM * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    W = lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget), wb.ctypes.data_as(C.c_void_p), 499)
    os.makedirs(d, exist_ok=True)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wb.ctypes.data_as(C.c_void_p), W) == 0


def run(cmd, cwd, log=None):
    t0 = time.time()
    with open(os.path.join(cwd, log), "w") if log else open(os.devnull, "w") as fh:
        subprocess.run(cmd, cwd=cwd, check=True, stdout=fh, stderr=subprocess.STDOUT)
    return time.time() - t0


def paint_targets(work):
    out = os.path.join(work, "pt16.bin")
    if not os.path.exists(out):
        run([rlutil.REF_HARNESS, "paint_targets", "out", "0", "pt16.bin"] + [str(k) for k in TARGETS], work, "pt16.log")


def paint_window(work, procs=7):
    if os.path.exists(os.path.join(work, "out", "chunk_0", "paint", "relate_%d.bin" % W_PIN)):
        return
    cuts = [N * i // procs for i in range(procs + 1)]
    ps = []
    for i in range(procs):
        part = os.path.join(work, "part%d_%d.bin" % (W_PIN, i))
        if os.path.exists(part + ".done"):
            continue
        ps.append((part, subprocess.Popen([rlutil.REF_HARNESS, "paint_window", "out", "0", str(W_PIN), str(cuts[i]),
                                           str(cuts[i + 1]), part], cwd=work, stderr=subprocess.DEVNULL)))
    for part, p in ps:
        assert p.wait() == 0
        open(part + ".done", "w").close()
    os.makedirs(os.path.join(work, "out", "chunk_0", "paint"), exist_ok=True)
    with open(os.path.join(work, "out", "chunk_0", "paint", "relate_%d.bin" % W_PIN), "wb") as fo:
        for i in range(procs):
            fo.write(open(os.path.join(work, "part%d_%d.bin" % (W_PIN, i)), "rb").read())


def build_topology(work):
    if not os.path.exists(os.path.join(work, "out", "chunk_0", "out_%d.anc" % W_PIN)):
        secs = run([rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", str(W_PIN),
                    "--last_section", str(W_PIN), "-o", "out"], work, "bt%d.log" % W_PIN)
        open(os.path.join(work, "bt%d.seconds" % W_PIN), "w").write("%.1f\n" % secs)


def matrix_snps(wb):
    s0, s1 = int(wb[W_PIN]), int(wb[W_PIN + 1]) - 1
    return [s0 + (s1 - s0) // 3, s1 - 1]


def matrices(work, wb):
    if not os.path.exists(os.path.join(work, "mx%d.bin" % W_PIN)):
        run([rlutil.REF_HARNESS, "matrix", "out", "0", str(W_PIN), "mx%d.bin" % W_PIN] + [str(s) for s in matrix_snps(wb)],
            work, "mx.log")
    if not os.path.exists(os.path.join(work, "rp%d.bin" % W_PIN)):
        run([rlutil.REF_HARNESS, "repaint_targets", "out", "0", str(W_PIN), "rp%d.bin" % W_PIN] +
            [str(k) for k in REPAINT_TARGETS], work, "rp.log")


def parse_record(b):
    """-> (start, end, bsnp_begin, ls_alpha, bsnp_end, ls_beta) of one paint-file record (collapsed_matrix.hpp:228-265)"""
    start, end = struct.unpack_from("<ii", b, 0)
    pos = 8
    out = [start, end]
    for _ in range(2):
        one, n, bsnp, ls, K = struct.unpack_from("<QQifi", b, pos)
        assert one == 1 and n == N
        pos += 28 + 8 * K
        out += [bsnp, ls]
    assert pos == len(b)
    return out


def pack(work):
    d = os.path.join(work, "out")
    data = {"mem": np.array([MEM]), "pin_window": np.array([W_PIN], dtype=np.int64)}
    for f in ["parameters_c0.bin", "chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.dist", "chunk_0.state"]:
        h = hashlib.md5()
        with open(os.path.join(d, f), "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
        data["in_md5/" + f] = np.frombuffer(h.digest(), dtype=np.uint8)
    p = np.fromfile(os.path.join(d, "parameters_c0.bin"), dtype=np.int32)
    W = int(p[2]) - 1
    wb = p[3:3 + W + 1].copy()
    data["wb"] = wb
    data["meta"] = np.array([N, L, W, SEED], dtype=np.int64)
    # step 2
    buf = open(os.path.join(work, "pt16.bin"), "rb").read()
    pos = 0
    rec_md5 = np.zeros((len(TARGETS), W, 16), dtype=np.uint8)
    rec_len = np.zeros((len(TARGETS), W), dtype=np.int32)
    ls = np.zeros((len(TARGETS), W, 2), dtype=np.float32)
    bs = np.zeros((len(TARGETS), W, 2), dtype=np.int32)
    for _ in range(len(TARGETS)):  # (in the order the harness was given them)
        for w in range(W):
            kk, ww, ln = struct.unpack_from("<iii", buf, pos)
            assert ww == w
            ti, k = TARGETS.index(kk), kk
            rec = buf[pos + 12: pos + 12 + ln]
            pos += 12 + ln
            rec_md5[ti, w] = md5(rec)
            rec_len[ti, w] = ln
            s, e, bb, la, be, lb = parse_record(rec)
            assert s == wb[w] and e == wb[w + 1] - 1
            ls[ti, w] = (la, lb)
            bs[ti, w] = (bb, be)
    assert pos == len(buf)
    data.update({"targets": np.array(TARGETS, dtype=np.int32), "record_md5": rec_md5, "record_len": rec_len,
                 "record_logscales": ls, "record_bsnp": bs})
    out = os.path.join(ROOT, "tests", "golden", "c3_full.npz")
    if not os.path.exists(os.path.join(d, "chunk_0", "out_%d.anc" % W_PIN)) or not os.path.exists(
            os.path.join(work, "rp%d.bin" % W_PIN)):  # (steps 3-5 not there yet: what there is)
        np.savez_compressed(out, **data)
        print("c3_full (records of %d targets only): %.1f KB" % (len(TARGETS), os.path.getsize(out) / 1e3))
        return
    # step 3
    pf = open(os.path.join(d, "chunk_0", "paint", "relate_%d.bin" % W_PIN), "rb").read()
    data["w/paint_md5"] = md5(pf)
    data["w/paint_size"] = np.array([len(pf)], dtype=np.int64)
    data["w/paint_head"] = np.frombuffer(pf[:1 << 16], dtype=np.uint8)
    # step 4
    anc = os.path.join(d, "chunk_0", "out_%d.anc" % W_PIN)
    mut = open(os.path.join(d, "chunk_0", "out_%d.mut" % W_PIN), "rb").read()
    h = hashlib.md5()
    with open(anc, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    data["w/anc_md5"] = np.frombuffer(h.digest(), dtype=np.uint8)
    data["w/anc_size"] = np.array([os.path.getsize(anc)], dtype=np.int64)
    data["w/mut_md5"] = md5(mut)
    data["w/mut"] = np.frombuffer(mut, dtype=np.uint8)
    _, trees = rlutil.parse_anc(anc)
    data["w/tree_pos"] = np.array([t[0] for t in trees], dtype=np.int32)
    data["w/tree_parent_md5"] = np.stack([md5(t[1].astype("<i4").tobytes()) for t in trees])
    for i in sorted(set([0, len(trees) // 2, len(trees) - 1])):
        data["w/tree_parent/%d" % i] = trees[i][1].astype(np.int32)
    try:
        data["w/reference_build_topology_s"] = np.array([float(open(os.path.join(work, "bt%d.seconds" % W_PIN)).read())])
    except Exception:
        pass
    # step 5
    snps = [int(wb[W_PIN])] + matrix_snps(wb)
    data["w/matrix_snps"] = np.array(snps, dtype=np.int32)
    data["w/matrix_rows_idx"] = np.array(MATRIX_ROWS, dtype=np.int32)
    with open(os.path.join(work, "mx%d.bin" % W_PIN), "rb") as fh:
        assert struct.unpack("<i", fh.read(4))[0] == N
        for i, s in enumerate(snps):
            assert struct.unpack("<i", fh.read(4))[0] == s
            m = np.frombuffer(fh.read(4 * N * N), dtype=np.float32).reshape(N, N)
            data["w/matrix_rows/%d" % i] = m[list(MATRIX_ROWS)].copy()
            data["w/matrix_md5/%d" % i] = md5(m.tobytes())
    with open(os.path.join(work, "rp%d.bin" % W_PIN), "rb") as fh:
        assert struct.unpack("<i", fh.read(4))[0] == N
        for k in REPAINT_TARGETS:
            assert struct.unpack("<i", fh.read(4))[0] == k
            D = struct.unpack("<i", fh.read(4))[0]
            data["w/repaint_logscales/%d" % k] = np.frombuffer(fh.read(4 * D), dtype=np.float32).copy()
            top = fh.read(4 * D * N)
            data["w/repaint_rows_md5/%d" % k] = md5(top)
            t = np.frombuffer(top, dtype=np.float32).reshape(D, N)
            data["w/repaint_row_first/%d" % k] = t[0].copy()
            data["w/repaint_row_last/%d" % k] = t[-1].copy()
    data["w/repaint_targets"] = np.array(REPAINT_TARGETS, dtype=np.int32)
    np.savez_compressed(out, **data)
    print("c3_full: W", W, "trees of section", W_PIN, len(trees), "%.1f KB" % (os.path.getsize(out) / 1e3))


if __name__ == "__main__":
    assert rlutil.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c3ref"
    steps = sys.argv[2:] or ["chunk", "paint_targets", "paint_window", "build_topology", "matrices", "pack"]
    os.makedirs(work, exist_ok=True)
    wb = None
    for s in steps:
        t0 = time.time()
        if s == "chunk":
            chunk(work)
        elif s == "paint_targets":
            paint_targets(work)
        elif s == "paint_window":
            paint_window(work)
        elif s == "build_topology":
            build_topology(work)
        elif s == "matrices":
            p = np.fromfile(os.path.join(work, "out", "parameters_c0.bin"), dtype=np.int32)
            matrices(work, p[3:3 + int(p[2])])
        elif s == "pack":
            pack(work)
        print("[make_golden_c3] %s: %.0f s" % (s, time.time() - t0), flush=True)
