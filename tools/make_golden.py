#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in a container that has /root/reference: it needs oracle/_ref/Relate
and oracle/_ref/ref_harness (`make -C oracle ref`).  The fixtures are data --
inputs written by this repo's generator and outputs produced by the unmodified
reference binary on them -- packed into compressed .npz files:

  synth24.npz        N=24, L=900 synthetic chunk, ~6 windows: chunk files, every
                     paint file, RePaintSection dump of one window, GetMatrix
                     dumps at several SNPs, per-section .anc/.mut, tree dump
  synth24_paint.npz  same chunk with --painting 0.025,2: paint files + matrices
  example8.npz       first 3000 SNPs of the reference's bundled example/data
                     (N=8) through MakeChunks (synthetic 1 cM/Mb map), Paint,
                     BuildTopology: BASELINE.json config #1 (plumbing)
  synth70.npz        N=70 (>64 lanes): paint files, matrices, .anc/.mut

Usage: python tools/make_golden.py
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def fbytes(path):
    return np.frombuffer(open(path, "rb").read(), dtype=np.uint8)


def run(cmd, cwd, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    subprocess.run(cmd, cwd=cwd, check=True, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def collect(work, out, W, painting=None, windows_dump=(1,), with_trees=True):
    """run the reference stages in `work` (chunk files already in work/out)"""
    pa = ["--painting", painting] if painting else []
    env = {"REF_PAINTING": painting} if painting else None
    run([rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", "0", "-o", out] + pa, work)
    data = {}
    for f in ["parameters_c0.bin", "chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.dist",
              "chunk_0.state"]:
        data["in/" + f] = fbytes(os.path.join(work, out, f))
    for w in range(W):
        data["paint/relate_%d.bin" % w] = fbytes(os.path.join(work, out, "chunk_0", "paint", "relate_%d.bin" % w))
    ch = rlutil.read_chunk(os.path.join(work, out))
    for w in windows_dump:
        run([rlutil.REF_HARNESS, "repaint", out, "0", str(w), "rp.bin"], work, env)
        data["repaint/w%d" % w] = fbytes(os.path.join(work, "rp.bin"))
        s0, s1 = int(ch.wb[w]), int(ch.wb[w + 1]) - 1
        snps = sorted(set([s0 + 1, s0 + (s1 - s0) // 2, s1]) - {s0})
        run([rlutil.REF_HARNESS, "matrix", out, "0", str(w), "mx.bin"] + [str(s) for s in snps], work, env)
        data["matrix/w%d" % w] = fbytes(os.path.join(work, "mx.bin"))
        data["matrix/w%d/snps" % w] = np.array([s0] + snps, dtype=np.int32)
    if with_trees:
        run([rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
             "--last_section", str(W - 1), "-o", out] + pa, work)
        for w in range(W):
            data["anc/%d" % w] = fbytes(os.path.join(work, out, "chunk_0", "%s_%d.anc" % (out, w)))
            data["mut/%d" % w] = fbytes(os.path.join(work, out, "chunk_0", "%s_%d.mut" % (out, w)))
        # the next stage downstream rewrites the .anc files in place (num_events, SNP_begin/end carried
        # along equivalent branches)
        run([rlutil.REF_RELATE, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", out], work)
        for w in range(W):
            data["feb_anc/%d" % w] = fbytes(os.path.join(work, out, "chunk_0", "%s_%d.anc" % (out, w)))
        # BuildTopology's options: --no_consistency, and --fb (a new tree at least every fb base pairs)
        for tag, opts in (("nc", ["--no_consistency"]), ("fb", ["--fb", "2500"])):
            run([rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                 "--last_section", str(W - 1), "-o", out] + pa + opts, work)
            for w in range(W):
                data["anc_%s/%d" % (tag, w)] = fbytes(os.path.join(work, out, "chunk_0", "%s_%d.anc" % (out, w)))
                data["mut_%s/%d" % (tag, w)] = fbytes(os.path.join(work, out, "chunk_0", "%s_%d.mut" % (out, w)))
    return data


def synth_fixture(name, N, L, seed, budget, painting=None, windows_dump=(1,), with_trees=True):
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    with tempfile.TemporaryDirectory() as work:
        ch.write(os.path.join(work, "out"))
        data = collect(work, "out", ch.W, painting, windows_dump, with_trees)
    data["meta"] = np.array([N, L, ch.W, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)
    print(name, "N", N, "L", L, "W", ch.W, "%.1f KB" % (os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1e3))


def noisy_fixture(name="synth40_noisy", N=40, L=600, seed=9, budget=9000):
    """A chunk the trees do NOT fit: every allele of the synthetic panel is flipped with probability 1 % and one SNP
    in 25 has its alleles swapped altogether -- SNPs that map only with the alleles flipped (.mut is_flipped = 1),
    SNPs that fit no branch and go onto several (ForceMapMutation, .mut is_mapping = 1), and many more trees per
    section than the clean fixtures have."""
    import ctypes as C
    from relate_amd import api
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=None)
    rng = np.random.RandomState(seed)
    seq = ch.seq.copy()
    flip = rng.rand(L, N) < 0.01
    flip[rng.rand(L) < 0.04] ^= True
    seq[flip] = ord("0") + ord("1") - seq[flip]
    wbuf = np.zeros(L + 2, dtype=np.int32)
    W = api.lib().rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget),
                                   wbuf.ctypes.data_as(C.c_void_p), L)
    assert W > 1
    ch = rlutil.Chunk(seq, ch.r, ch.rpos, wbuf[:W + 1].copy(), ch.bp)
    with tempfile.TemporaryDirectory() as work:
        ch.write(os.path.join(work, "out"))
        data = collect(work, "out", ch.W, None, windows_dump=(), with_trees=True)
    data["meta"] = np.array([N, L, ch.W, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)
    multi = flipped = total = 0
    for w in range(ch.W):
        for line in data["mut/%d" % w].tobytes().decode().split("\n")[1:]:
            f = line.split(";")
            if len(f) > 3:
                total += 1
                multi += f[2] == "1"
                flipped += f[3] == "1"
    print(name, "N", N, "L", L, "W", ch.W, "SNPs", total, "on several branches", multi, "flipped", flipped,
          "%.1f KB" % (os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1e3))


def example_fixture(nsnps=3000):
    src = "/root/reference/example/data"
    with tempfile.TemporaryDirectory() as work:
        lines = []
        with gzip.open(os.path.join(src, "example.haps.gz"), "rt") as f:
            for i, line in enumerate(f):
                if i >= nsnps:
                    break
                lines.append(line)
        open(os.path.join(work, "ex.haps"), "w").writelines(lines)
        with gzip.open(os.path.join(src, "example.sample.gz"), "rt") as f:
            open(os.path.join(work, "ex.sample"), "w").write(f.read())
        first = int(lines[0].split()[2])
        last = int(lines[-1].split()[2])
        # synthetic uniform 1 cM/Mb map (the bundled map is a missing blob): docs/input_data.html
        with open(os.path.join(work, "ex.map"), "w") as f:
            f.write("pos COMBINED_rate Genetic_Map\n")
            for bp in range(max(0, first - 50000), last + 100000, 50000):
                f.write("%d 1.0 %.6f\n" % (bp, bp * 1e-6))
        run([rlutil.REF_RELATE, "--mode", "MakeChunks", "--haps", "ex.haps", "--sample", "ex.sample", "--map",
             "ex.map", "--memory", "0.0002", "-o", "example"], work)
        p = np.fromfile(os.path.join(work, "example", "parameters_c0.bin"), dtype=np.int32)
        W = int(p[2]) - 1
        data = collect(work, "example", W, None, windows_dump=(0,), with_trees=True)
        data["meta"] = np.array([p[0], p[1], W, 0], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "example8.npz"), **data)
    print("example8 N", p[0], "L", p[1], "W", W, "%.1f KB" % (os.path.getsize(os.path.join(GOLD, "example8.npz")) / 1e3))


def n5000_fixture(N=5000, L=1200, mem=10.0, name="n5000"):
    """(name = "n10000", N = 10000, L = 600, mem = 12: BASELINE.json config #5's N -- two wavefronts per target in K1 /
    K2, the device tree builder's global-memory state path -- about an hour of the single-threaded reference.)
    The headline tile (BASELINE.json config #3's N = 5000, one wavefront of S = 80 registers per target) against
    the reference binary: a short chunk (3 windows), Paint of the whole chunk and BuildTopology of section 0.
    The inputs are regenerated from the seed by the test (their md5s are kept); of the outputs the fixture
    keeps the md5 of every file, the head of window 0's paint file and the parent arrays of the section's trees
    (md5 per tree, the first and the last two in full)."""
    import ctypes as C
    import hashlib
    from relate_amd import api
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
    assert W >= 2, W
    md5 = lambda b: np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)
    data = {"meta": np.array([N, L, W, 1], dtype=np.int64), "mem": np.array([mem]), "wb": wb[:W + 1].copy()}
    with tempfile.TemporaryDirectory() as work:
        d = os.path.join(work, "out")
        os.makedirs(d)
        lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p),
                                        bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                        rpos.ctypes.data_as(C.c_void_p), wb.ctypes.data_as(C.c_void_p), W) == 0
        for f in ["parameters_c0.bin", "chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.dist",
                  "chunk_0.state"]:
            data["in_md5/" + f] = md5(open(os.path.join(d, f), "rb").read())
        run([rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], work)
        for w in range(W):
            b = open(os.path.join(d, "chunk_0", "paint", "relate_%d.bin" % w), "rb").read()
            data["md5/paint/relate_%d.bin" % w] = md5(b)
            if w == 0:
                data["head/paint/relate_0.bin"] = np.frombuffer(b[:1 << 16], dtype=np.uint8)
        run([rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
             "--last_section", "0", "-o", "out"], work)
        anc = open(os.path.join(d, "chunk_0", "out_0.anc"), "rb").read()
        mut = open(os.path.join(d, "chunk_0", "out_0.mut"), "rb").read()
        data["md5/out_0.anc"] = md5(anc)
        data["md5/out_0.mut"] = md5(mut)
        data["mut/0"] = np.frombuffer(mut, dtype=np.uint8)
        _, trees = rlutil.parse_anc(os.path.join(d, "chunk_0", "out_0.anc"))
        data["tree_pos"] = np.array([t[0] for t in trees], dtype=np.int32)
        data["tree_parent_md5"] = np.stack([md5(t[1].astype("<i4").tobytes()) for t in trees])
        for i in sorted(set([0, len(trees) - 2, len(trees) - 1])):
            data["tree_parent/%d" % i] = trees[i][1].astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)
    print(name, "N", N, "L", L, "W", W, "trees", len(trees),
          "%.1f KB" % (os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1e3))


def n5000_matrix_fixture(N=5000, L=1200, mem=10.0, rows=(0, 1, 2, 1250, 2500, 3750, 4998, 4999)):
    """GetMatrix of the reference at the headline tile (the chunk of n5000_fixture): window 0's distance matrix at its
    first SNP and at two later ones -- `rows` of each in full (a 5000 x 5000 matrix is 100 MB), the md5 of the whole
    matrix, and the largest |logscale| of the window's posterior rows (the scale of the tolerance SURVEY.md 7 H1
    derives for re-associated sums)."""
    import ctypes as C
    import hashlib
    from relate_amd import api
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
    assert W >= 2, W
    s0, s1 = int(wb[0]), int(wb[1]) - 1
    snps = [s0 + (s1 - s0) // 3, s1 - 1]
    data = {"meta": np.array([N, L, W, 1], dtype=np.int64), "mem": np.array([mem]), "rows": np.array(rows, dtype=np.int32),
            "snps": np.array([s0] + snps, dtype=np.int32)}
    with tempfile.TemporaryDirectory() as work:
        d = os.path.join(work, "out")
        os.makedirs(d)
        lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p),
                                        bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                        rpos.ctypes.data_as(C.c_void_p), wb.ctypes.data_as(C.c_void_p), W) == 0
        run([rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], work)
        run([rlutil.REF_HARNESS, "matrix", "out", "0", "0", "mx.bin"] + [str(s) for s in snps], work)
        buf = open(os.path.join(work, "mx.bin"), "rb").read()
        assert np.frombuffer(buf, np.int32, 1, 0)[0] == N
        pos = 4
        for i in range(3):
            s = int(np.frombuffer(buf, np.int32, 1, pos)[0]); pos += 4
            assert s == data["snps"][i]
            m = np.frombuffer(buf, np.float32, N * N, pos).reshape(N, N); pos += 4 * N * N
            data["matrix_rows/%d" % i] = m[list(rows)].copy()
            data["matrix_md5/%d" % i] = np.frombuffer(hashlib.md5(m.tobytes()).digest(), dtype=np.uint8)
        # the window's logscales: RePaintSection dump of a few targets (each D x N floats) is enough for the scale
        run([rlutil.REF_HARNESS, "repaint", "out", "0", "0", "rp.bin"], work)
        rb = open(os.path.join(work, "rp.bin"), "rb")
        assert np.frombuffer(rb.read(4), np.int32)[0] == N
        scale = 0.0
        for n in range(N):
            D = int(np.frombuffer(rb.read(4), np.int32)[0])
            ls = np.frombuffer(rb.read(4 * D), np.float32)
            scale = max(scale, float(np.abs(ls).max()))
            rb.seek(4 * D * N, 1)
        data["logscale_max"] = np.array([scale])
    np.savez_compressed(os.path.join(GOLD, "n5000_matrix.npz"), **data)
    print("n5000_matrix snps", data["snps"], "max |logscale| %.1f" % scale,
          "%.1f KB" % (os.path.getsize(os.path.join(GOLD, "n5000_matrix.npz")) / 1e3))


def pipeline_fixture(N=6, L=50000, memory="0.0005"):
    """The many-chunks route end to end (BASELINE.json config #4's shape, scripts/RelateParallel/RelateParallel.sh:216-262)
    on the synthetic .haps of tests/test_makechunks.py: the reference's MakeChunks (3 overlapping chunks), then per
    chunk Paint, BuildTopology of all sections and FindEquivalentBranches.  The fixture keeps the md5 of every paint
    file and of every .anc / .mut as FindEquivalentBranches leaves them; the test regenerates the inputs from the seed."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_makechunks import write_synth_haps
    md5 = lambda b: np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)
    data = {"args": np.array([N, L], dtype=np.int64), "memory": np.array([float(memory)])}
    with tempfile.TemporaryDirectory() as work:
        write_synth_haps(work, N, L, seed=N)
        for fn in ("s.haps", "s.sample", "s.map"):
            data["in_md5/" + fn] = md5(open(os.path.join(work, fn), "rb").read())
        run([rlutil.REF_RELATE, "--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map",
             "--memory", memory, "-o", "job"], work)
        C = int(np.fromfile(os.path.join(work, "job", "parameters.bin"), dtype=np.int32, count=3)[2])
        data["num_chunks"] = np.array([C], dtype=np.int64)
        total = 0
        for c in range(C):
            W = int(np.fromfile(os.path.join(work, "job", "parameters_c%d.bin" % c), dtype=np.int32, count=3)[2]) - 1
            run([rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", str(c), "-o", "job"], work)
            for w in range(W):
                data["c%d/paint/relate_%d.bin" % (c, w)] = md5(
                    open(os.path.join(work, "job", "chunk_%d" % c, "paint", "relate_%d.bin" % w), "rb").read())
            run([rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", str(c), "--first_section", "0",
                 "--last_section", str(W - 1), "-o", "job"], work)
            run([rlutil.REF_RELATE, "--mode", "FindEquivalentBranches", "--chunk_index", str(c), "-o", "job"], work)
            for w in range(W):
                for ext in ("anc", "mut"):
                    data["c%d/job_%d.%s" % (c, w, ext)] = md5(
                        open(os.path.join(work, "job", "chunk_%d" % c, "job_%d.%s" % (w, ext)), "rb").read())
            data["c%d/sections" % c] = np.array([W], dtype=np.int64)
            total += W
    np.savez_compressed(os.path.join(GOLD, "pipeline6.npz"), **data)
    print("pipeline6: %d chunks, %d sections, %.1f KB" % (C, total, os.path.getsize(os.path.join(GOLD, "pipeline6.npz")) / 1e3))


def ages_fixture(name="synth24_ages", N=24, L=900, seed=21, budget=4000):
    """BuildTopology --sample_ages (ancient samples) of the synth24 chunk: the ages file and the reference's .anc / .mut
    of every section, with the default consistency prior and with --no_consistency"""
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=budget)
    rng = np.random.RandomState(seed)
    ages = np.zeros(N)
    ages[rng.permutation(N)[:5]] = [300.0, 300.0, 1200.0, 1200.0, 4000.0]
    data = {"meta": np.array([N, L, ch.W, seed], dtype=np.int64), "ages": ages}
    with tempfile.TemporaryDirectory() as work:
        ch.write(os.path.join(work, "out"))
        with open(os.path.join(work, "ages.txt"), "w") as f:
            f.write("\n".join("%g" % a for a in ages) + "\n")
        run([rlutil.REF_RELATE, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], work)
        for f in ["parameters_c0.bin", "chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.dist",
                  "chunk_0.state"]:
            data["in/" + f] = fbytes(os.path.join(work, "out", f))
        for w in range(ch.W):
            data["paint/relate_%d.bin" % w] = fbytes(os.path.join(work, "out", "chunk_0", "paint", "relate_%d.bin" % w))
        for tag, opts in (("", []), ("_nc", ["--no_consistency"])):
            run([rlutil.REF_RELATE, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0",
                 "--last_section", str(ch.W - 1), "--sample_ages", "ages.txt", "-o", "out"] + opts, work)
            for w in range(ch.W):
                data["anc%s/%d" % (tag, w)] = fbytes(os.path.join(work, "out", "chunk_0", "out_%d.anc" % w))
                data["mut%s/%d" % (tag, w)] = fbytes(os.path.join(work, "out", "chunk_0", "out_%d.mut" % w))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)
    print(name, "W", ch.W, "%.1f KB" % (os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1e3))


def makechunks_fixture():
    """MakeChunks off the build container: the synthetic .haps/.sample/map of tests/test_makechunks.py (regenerated
    from the seed by the test) through the reference's MakeChunks; the fixture keeps the md5 of every output file and
    the parameter files in full (multi-chunk case with the 20000-SNP overlap, and --transversion)."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_makechunks import write_synth_haps
    data = {}
    # (c: an allowance whose window budget, memory * 1e9 / 4 floats, is not a whole number)
    for tag, N, L, memory, extra in (("a", 6, 50000, "0.0005", []), ("b", 8, 3000, "0.0002", ["--transversion"]),
                                     ("c", 8, 3000, "0.00020001", [])):
        with tempfile.TemporaryDirectory() as work:
            write_synth_haps(work, N, L, seed=N)
            for fn in ("s.haps", "s.sample", "s.map"):
                data["%s/in_md5/%s" % (tag, fn)] = np.frombuffer(
                    hashlib.md5(open(os.path.join(work, fn), "rb").read()).digest(), dtype=np.uint8)
            run([rlutil.REF_RELATE, "--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map",
                 "s.map", "--memory", memory] + extra + ["-o", "ref"], work)
            for fn in sorted(os.listdir(os.path.join(work, "ref"))):
                b = open(os.path.join(work, "ref", fn), "rb").read()
                data["%s/md5/%s" % (tag, fn)] = np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)
                if fn.startswith("parameters"):
                    data["%s/file/%s" % (tag, fn)] = np.frombuffer(b, dtype=np.uint8)
        data["%s/args" % tag] = np.array([N, L], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "makechunks.npz"), **data)
    print("makechunks fixture: %d entries, %.1f KB" % (len(data), os.path.getsize(os.path.join(GOLD, "makechunks.npz")) / 1e3))


def builder_fixture():
    """tests/golden/builder_adversarial.npz: the parent arrays the reference's MinMatch (one object per sequence,
    oracle/ref_harness quickbuild_seq) gives for the sequences of tests/builder_cases.py"""
    import builder_cases
    data = {}
    for name, make in builder_cases.CASES.items():
        N, mats = make()
        work = tempfile.mkdtemp()
        try:
            args = [rlutil.REF_HARNESS, "quickbuild_seq", str(N), os.path.join(work, "p.bin")]
            for t, (d, prior) in enumerate(mats):
                d.tofile(os.path.join(work, "d%d.bin" % t))
                args.append(os.path.join(work, "d%d.bin" % t))
                if prior is None:
                    args.append("-")
                else:
                    prior.tofile(os.path.join(work, "c%d.bin" % t))
                    args.append(os.path.join(work, "c%d.bin" % t))
            subprocess.run(args, check=True)
            data[name] = np.fromfile(os.path.join(work, "p.bin"), dtype=np.int32).reshape(len(mats), 2 * N - 1)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(GOLD, "builder_adversarial.npz"), **data)
    print("builder_adversarial", {k: v.shape for k, v in data.items()}, "%.1f KB" % (os.path.getsize(os.path.join(GOLD, "builder_adversarial.npz")) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "builder":
        builder_fixture()
        sys.exit(0)
    assert rlutil.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "ages":
        ages_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pipeline":
        pipeline_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "noisy":
        noisy_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "makechunks":
        makechunks_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n5000_matrix":  # a few minutes, ~30 GB of scratch files
        n5000_matrix_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n10000":  # ~1 h of the single-threaded reference, ~15 GB of memory
        n5000_fixture(N=10000, L=600, mem=12.0, name="n10000")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n5000":  # ~15 minutes of the single-threaded reference
        n5000_fixture()
        sys.exit(0)
    synth_fixture("synth24", 24, 900, seed=21, budget=4000)
    synth_fixture("synth24_paint", 24, 900, seed=21, budget=4000, painting="0.025,2", with_trees=False)
    synth_fixture("synth70", 70, 700, seed=5, budget=40000, windows_dump=(0, 2))
    example_fixture()
