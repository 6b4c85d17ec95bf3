"""The HEADLINE configuration against the unmodified reference at FULL length (BASELINE.json config #3: synthetic
N = 5000 x L = 500,000, seed 1, --memory 20, 267 windows).  tests/golden/c3_full.npz
(tools/make_golden_c3.py, hours of the single-threaded reference in the build container) holds

  * FastPainting::PaintSteppingStones (fast_painting.cpp:18-618) at full length for 16 targets: the md5 of every
    (window, target) record of the paint files and both logscales of every record -- D_k ~ 55,000 dependent steps
    per target, ~500 rescalings, |logscale| in the thousands;
  * the reference's COMPLETE paint file of window 133 (all 5000 targets) by md5, size and head;
  * `Relate --mode BuildTopology` of section 133 on that file (pipeline/BuildTopology.cpp:125-150): md5 of .anc and
    .mut, the .mut in full, md5 of every tree's parent array, three arrays in full;
  * DistanceMeasure::GetMatrix of window 133 at three SNPs (md5 + 8 rows) and RePaintSection of four targets.

Here: ONE Paint of the whole chunk on the device (1.4 s) and everything above compared bit for bit; 16 more
(seeded) targets against the oracle at full length; the fused stage for section 133 with both tree builders, the
device builder with bounded windows as the whole-chunk stage runs them.

tests/golden/c3_ends.npz (tools/make_golden_full.py c3_ends; round 6): the two BOUNDARY windows, 0
(boundarySNP_begin = 0) and 266 (boundarySNP_end = L - 1, the last SNP's no-carrier quirk) -- the windows with
special-case code in fast_painting.cpp:60-69, :98-107, :150 -- as the reference wrote them: its complete paint file
of each and BuildTopology of each section."""
import ctypes as C
import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import rlutil
from bigtile import md5
from relate_amd import api

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "c3_full.npz")


def synth(z):
    """the seed-1 C3 panel bit-packed + the reference's window rule (what bench.py paints)"""
    N, L, W, seed = [int(x) for x in z["meta"]]
    lib = api.lib()
    rw = (N + 31) // 32
    bits = np.zeros((L, rw), dtype=np.uint32)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, None, bits.ctypes.data_as(C.c_void_p), rw,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = float(z["mem"][0]) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows_bits(N, L, bits.ctypes.data_as(C.c_void_p), rw, C.c_double(budget),
                                     wb.ctypes.data_as(C.c_void_p), 499) == W
    wb = wb[:W + 1].copy()
    assert np.array_equal(wb, z["wb"])
    return bits, bp, r, rpos, wb


@pytest.fixture(scope="module")
def c3():
    z = np.load(GOLD)
    bits, bp, r, rpos, wb = synth(z)
    N = int(z["meta"][0])
    ctx = api.Context()
    ctx.set_chunk_bits(N, bits, r, rpos, wb)
    ctx.prepare()
    ms = ctx.paint(api.RL_SUM_EXACT)
    assert (ctx.N, ctx.L, ctx.W, ctx.tile, ctx.waves) == (5000, 500000, 267, 80, 1)
    print("C3 Paint (exact): %.1f ms" % ms)
    yield z, ctx, (bits, bp, r, rpos, wb)
    ctx.close()


def record_logscales(rec, N):
    start, end = struct.unpack_from("<ii", rec, 0)
    pos, out = 8, []
    for _ in range(2):
        one, n, bsnp, ls, K = struct.unpack_from("<QQifi", rec, pos)
        assert one == 1 and n == N
        out += [bsnp, ls]
        pos += 28 + 8 * K
    assert pos == len(rec)
    return start, end, out


def test_records_of_16_targets_are_the_references_at_full_length(c3):
    """every (window, target) record of 16 targets: the bytes PaintSteppingStones appends to pfiles[w]"""
    z, ctx, _ = c3
    N, W = ctx.N, ctx.W
    bad = []
    for ti, k in enumerate(int(x) for x in z["targets"]):
        for w in range(W):
            rec = ctx.paint_record(w, k)
            if len(rec) != int(z["record_len"][ti, w]) or not np.array_equal(md5(rec), z["record_md5"][ti, w]):
                s, e, (bb, la, be, lb) = record_logscales(rec, N)
                bad.append((k, w, bb, la, be, lb, [float(x) for x in z["record_logscales"][ti, w]],
                            [int(x) for x in z["record_bsnp"][ti, w]]))
    assert not bad, "%d of %d records differ from the reference's; first: %s" % (len(bad), 16 * W, bad[:3])
    # the scale of what was compared: the logscales run into the thousands at this length
    assert float(np.abs(z["record_logscales"]).max()) > 1000.0


def test_16_more_targets_against_the_oracle_at_full_length(c3):
    z, ctx, (bits, bp, r, rpos, wb) = c3
    N, L, W = ctx.N, ctx.L, ctx.W
    o = rlutil.oracle()
    seq = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :N]
    seq = np.ascontiguousarray(seq + ord("0"), dtype=np.uint8)
    d = rlutil.RoData(N, L, seq.ctypes.data, r.ctypes.data, rpos.ctypes.data, 0.001)
    rng = np.random.RandomState(2026)
    pinned = set(int(x) for x in z["targets"])
    targets = [int(k) for k in rng.permutation(N) if int(k) not in pinned][:16]
    order = rlutil.RoSumOrder(0, 0, 0)  # RO_SUM_SERIAL: the reference's order
    o.ro_encode_stone.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    cap = o.ro_stone_max_bytes(N)

    def one(k):
        a = np.zeros((W, N), np.float32)
        b = np.zeros((W, N), np.float32)
        la = np.zeros(W, np.float32)
        lb = np.zeros(W, np.float32)
        bb = np.zeros(W, np.int32)
        be = np.zeros(W, np.int32)
        rc = o.ro_paint_stepping_stones(C.byref(d), wb.ctypes.data_as(C.c_void_p), W, k, C.byref(order),
                                        bb.ctypes.data_as(C.c_void_p), be.ctypes.data_as(C.c_void_p),
                                        a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                        la.ctypes.data_as(C.c_void_p), lb.ctypes.data_as(C.c_void_p))
        assert rc > 0  # (the sites visited)
        recs = []
        buf = C.create_string_buffer(cap)
        for w in range(W):
            rec = struct.pack("<ii", int(wb[w]), int(wb[w + 1]) - 1)
            n1 = o.ro_encode_stone(a[w].ctypes.data_as(C.c_void_p), N, int(bb[w]), float(la[w]), buf)
            rec += buf.raw[:n1]
            n2 = o.ro_encode_stone(b[w].ctypes.data_as(C.c_void_p), N, int(be[w]), float(lb[w]), buf)
            rec += buf.raw[:n2]
            recs.append(rec)
        return recs

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        want = list(pool.map(one, targets))
    for k, recs in zip(targets, want):
        for w in range(W):
            assert ctx.paint_record(w, k) == recs[w], "record of target %d, window %d differs from the oracle's" % (k, w)


def test_paint_file_of_one_window_is_the_references(c3, tmp_path):
    z, ctx, _ = c3
    w = int(z["pin_window"][0])
    fn = str(tmp_path / "relate_w.bin")
    ctx.write_paint_file(w, fn)
    b = open(fn, "rb").read()
    head = z["w/paint_head"].tobytes()
    assert b[:len(head)] == head
    assert len(b) == int(z["w/paint_size"][0])
    assert np.array_equal(md5(b), z["w/paint_md5"])


@pytest.mark.parametrize("route", ["paint_file", "stones_in_hbm"])
def test_repaint_and_matrices_of_that_window_are_the_references(c3, tmp_path, route):
    """RePaintSection of four targets (all posterior rows by md5, logscales, first and last row in full) and
    GetMatrix at three SNPs (md5 of the 5000 x 5000 matrix, 8 rows in full), from the paint file and from the
    stones left in HBM with the file's quantisation applied on the device (the fused stage's route)"""
    z, ctx, _ = c3
    w = int(z["pin_window"][0])
    pf = None
    if route == "paint_file":
        pf = str(tmp_path / "relate_w.bin")
        ctx.write_paint_file(w, pf)
    snps = [int(x) for x in z["w/matrix_snps"]]
    win = ctx.open_window(w, pf, snps[0])
    for k in (int(x) for x in z["w/repaint_targets"]):
        top, ls = win.topology(k)
        assert np.array_equal(ls.view(np.uint32), z["w/repaint_logscales/%d" % k].view(np.uint32)), k
        assert np.array_equal(top[0].view(np.uint32), z["w/repaint_row_first/%d" % k].view(np.uint32)), k
        assert np.array_equal(top[-1].view(np.uint32), z["w/repaint_row_last/%d" % k].view(np.uint32)), k
        assert np.array_equal(md5(np.ascontiguousarray(top).tobytes()), z["w/repaint_rows_md5/%d" % k]), k
    rows = [int(x) for x in z["w/matrix_rows_idx"]]
    cur = snps[0]
    for i, s in enumerate(snps):
        for t in range(cur + 1, s + 1):
            win.advance(t)
        cur = s
        g = win.matrix(s)
        assert np.array_equal(g[rows].view(np.uint32), z["w/matrix_rows/%d" % i].view(np.uint32)), s
        assert np.array_equal(md5(np.ascontiguousarray(g).tobytes()), z["w/matrix_md5/%d" % i]), s
    win.close()


@pytest.fixture(scope="module")
def chunk_dir(c3, tmp_path_factory):
    """the chunk files the reference was given (md5-checked)"""
    z, ctx, (bits, bp, r, rpos, wb) = c3
    N, L, W = ctx.N, ctx.L, ctx.W
    work = str(tmp_path_factory.mktemp("c3full"))
    d = os.path.join(work, "out")
    os.makedirs(d)
    seq = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :N]
    seq = np.ascontiguousarray(seq + ord("0"), dtype=np.uint8)
    lib = api.lib()
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    wbf = np.zeros(L + 2, dtype=np.int32)
    wbf[:W + 1] = wb
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                    r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                    wbf.ctypes.data_as(C.c_void_p), W) == 0
    del seq
    import hashlib
    for key in z.files:
        if key.startswith("in_md5/"):
            h = hashlib.md5()
            with open(os.path.join(d, key[7:]), "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk)
            assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), z[key]), key
    return work


def check_section(z, out_dir, w):
    anc = os.path.join(out_dir, "chunk_0", "out_%d.anc" % w)
    mut = open(os.path.join(out_dir, "chunk_0", "out_%d.mut" % w), "rb").read()
    _, trees = rlutil.parse_anc(anc)
    assert [t[0] for t in trees] == list(z["w/tree_pos"]), "tree positions"
    for t, (tr, want) in enumerate(zip(trees, z["w/tree_parent_md5"])):
        if "w/tree_parent/%d" % t in z.files:
            assert np.array_equal(tr[1], z["w/tree_parent/%d" % t]), "parent array of tree %d" % t
        assert np.array_equal(md5(tr[1].astype("<i4").tobytes()), want), "parent array of tree %d" % t
    assert mut == z["w/mut"].tobytes()
    assert np.array_equal(md5(mut), z["w/mut_md5"])
    assert os.path.getsize(anc) == int(z["w/anc_size"][0])
    assert np.array_equal(md5(open(anc, "rb").read()), z["w/anc_md5"])
    os.remove(anc)
    os.remove(os.path.join(out_dir, "chunk_0", "out_%d.mut" % w))


@pytest.mark.parametrize("builder", ["host", "gpu_bounded"])
def test_fused_stage_of_that_section_writes_the_references_files(c3, chunk_dir, builder):
    """Relate --mode PaintBuildTopology for section 133 alone (Paint of the whole chunk, stones in HBM): the reference's
    out_133.anc / out_133.mut -- trees by the host's MinMatch with the whole window resident, and by the device's
    workers with 1/32 of the window's rows resident (the whole-chunk stage's schedule: ~37 RePaint launches)"""
    z, ctx, _ = c3
    w = int(z["pin_window"][0])
    rows = 0
    if builder == "gpu_bounded":
        win = ctx.open_window(w, None, int(z["wb"][w]))
        rows = sum(win.rows(n) for n in range(ctx.N)) // 32
        win.close()
    opts = api.stage_opts(gpu_build=0 if builder == "host" else 1, window_rows=rows if rows else -1)
    api.stage_build_topology_ex(os.path.join(chunk_dir, "out"), 0, w, w, opts, fused=True)
    check_section(z, os.path.join(chunk_dir, "out"), w)


ENDS = os.path.join(ROOT, "tests", "golden", "c3_ends.npz")


@pytest.fixture(scope="module")
def ends(c3):
    z = np.load(ENDS)
    zc, ctx, _ = c3
    assert [int(x) for x in z["meta"]] == [int(x) for x in zc["meta"]] and np.array_equal(z["wb"], zc["wb"])
    for key in z.files:  # (the same chunk files as c3_full's reference runs were given)
        if key.startswith("in_md5/"):
            assert np.array_equal(z[key], zc[key]), key
    return z


@pytest.mark.parametrize("which", [0, 1])
def test_paint_files_of_the_boundary_windows_are_the_references(c3, ends, tmp_path, which):
    _, ctx, _ = c3
    w = int(ends["sections"][which])
    assert w in (0, ctx.W - 1)
    fn = str(tmp_path / "relate_w.bin")
    ctx.write_paint_file(w, fn)
    assert os.path.getsize(fn) == int(ends["s%d/paint_size" % w][0])
    assert np.array_equal(md5(open(fn, "rb").read()), ends["s%d/paint_md5" % w])


def check_end_section(z, out_dir, w):
    anc = os.path.join(out_dir, "chunk_0", "out_%d.anc" % w)
    mut = open(os.path.join(out_dir, "chunk_0", "out_%d.mut" % w), "rb").read()
    _, trees = rlutil.parse_anc(anc)
    assert [t[0] for t in trees] == list(z["s%d/tree_pos" % w]), "tree positions"
    for t, (tr, want) in enumerate(zip(trees, z["s%d/tree_parent_md5" % w])):
        assert np.array_equal(md5(tr[1].astype("<i4").tobytes()), want), "parent array of tree %d" % t
    assert mut == z["s%d/mut" % w].tobytes()
    assert os.path.getsize(anc) == int(z["s%d/anc_size" % w][0])
    assert np.array_equal(md5(open(anc, "rb").read()), z["s%d/anc_md5" % w])
    os.remove(anc)
    os.remove(os.path.join(out_dir, "chunk_0", "out_%d.mut" % w))


@pytest.mark.parametrize("which", [0, 1])
def test_fused_stage_of_the_boundary_sections_writes_the_references_files(c3, ends, chunk_dir, which):
    """Relate --mode PaintBuildTopology for section 0 / 266 alone, the device's workers, 1/32 of the window's rows
    resident: the reference's out_<w>.anc / out_<w>.mut"""
    _, ctx, _ = c3
    w = int(ends["sections"][which])
    win = ctx.open_window(w, None, int(ends["wb"][w]))
    rows = sum(win.rows(n) for n in range(ctx.N)) // 32
    win.close()
    opts = api.stage_opts(gpu_build=1, window_rows=rows)
    api.stage_build_topology_ex(os.path.join(chunk_dir, "out"), 0, w, w, opts, fused=True)
    check_end_section(ends, os.path.join(chunk_dir, "out"), w)
