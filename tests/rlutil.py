"""Shared helpers for the tests, bench.py and tools/: ctypes bindings of the
oracle (TEST INFRASTRUCTURE) and small file-format parsers.

Nothing in relate_amd/ imports this module.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_RELATE = os.path.join(ORACLE_DIR, "_ref", "Relate")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")


class RoData(C.Structure):
    _fields_ = [("N", C.c_int), ("L", C.c_int), ("seq", C.c_void_p),
                ("r", C.c_void_p), ("rpos", C.c_void_p), ("theta", C.c_double)]


class RoSumOrder(C.Structure):
    _fields_ = [("mode", C.c_int), ("seg", C.c_int), ("nwaves", C.c_int)]


_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def oracle():
    """ctypes handle of oracle/liboracle.so (built on demand)."""
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.ro_fast_log.restype = C.c_float
        lib.ro_fast_log.argtypes = [C.c_float]
        lib.ro_paint_stepping_stones.restype = C.c_int
        lib.ro_paint_chunk.restype = C.c_int
        lib.ro_paint_sample.restype = C.c_longlong
        lib.ro_repaint_section.restype = C.c_int
        lib.ro_window_open.restype = C.c_void_p
        lib.ro_window_top.restype = C.c_void_p
        lib.ro_window_log.restype = C.c_void_p
        lib.ro_stone_max_bytes.restype = C.c_size_t
        lib.ro_encode_stone.restype = C.c_size_t
        lib.ro_decode_stone.restype = C.c_size_t
        _oracle = lib
    return _oracle


class Chunk:
    """One chunk in host memory (numpy) + the oracle view of it."""

    def __init__(self, seq, r, rpos, wb, bp=None, theta=0.001):
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)  # L x N chars
        self.L, self.N = self.seq.shape
        self.r = np.ascontiguousarray(r, dtype=np.float64)
        self.rpos = np.ascontiguousarray(rpos, dtype=np.float64)
        self.wb = np.ascontiguousarray(wb, dtype=np.int32)
        self.W = len(self.wb) - 1
        self.bp = (np.ascontiguousarray(bp, dtype=np.int32) if bp is not None
                   else np.arange(self.L, dtype=np.int32) * 100 + 1000)
        self.theta = theta

    def ro(self):
        d = RoData(self.N, self.L, self.seq.ctypes.data, self.r.ctypes.data,
                   self.rpos.ctypes.data, self.theta)
        return d

    def bits(self, row_words=None):
        """bit-packed panel: uint32 [L][row_words], bit n of row s = derived"""
        rw = row_words or (self.N + 31) // 32
        b = np.zeros((self.L, rw * 32), dtype=np.uint8)
        b[:, :self.N] = (self.seq == ord("1"))
        packed = np.packbits(b, axis=1, bitorder="little")
        return np.ascontiguousarray(packed).view(np.uint32).reshape(self.L, rw)

    def write(self, out_dir, chunk=0):
        os.makedirs(out_dir, exist_ok=True)
        b = os.path.join(out_dir, "chunk_%d" % chunk)
        with open(b + ".hap", "wb") as f:
            f.write(struct.pack("<QQ", self.L, self.N))
            f.write(self.seq.tobytes())
        dist = np.ones(self.L, dtype=np.int32)
        dist[:-1] = np.diff(self.bp)
        for ext, hdr, arr in ((".bp", self.L, self.bp), (".dist", self.L, dist),
                              (".r", self.L, self.r), (".rpos", self.L + 1, self.rpos),
                              (".state", self.L, np.ones(self.L, dtype=np.int32))):
            with open(b + ext, "wb") as f:
                f.write(struct.pack("<I", hdr))
                f.write(arr.tobytes())
        with open(os.path.join(out_dir, "parameters_c%d.bin" % chunk), "wb") as f:
            f.write(struct.pack("<iii", self.N, self.L, self.W + 1))
            f.write(self.wb.tobytes())


def read_chunk(out_dir, chunk=0, theta=0.001):
    b = os.path.join(out_dir, "chunk_%d" % chunk)
    with open(b + ".hap", "rb") as f:
        L, N = struct.unpack("<QQ", f.read(16))
        seq = np.frombuffer(f.read(L * N), dtype=np.uint8).reshape(L, N)
    r = np.fromfile(b + ".r", dtype=np.float64, offset=4)
    rpos = np.fromfile(b + ".rpos", dtype=np.float64, offset=4)
    bp = np.fromfile(b + ".bp", dtype=np.int32, offset=4)
    p = np.fromfile(os.path.join(out_dir, "parameters_c%d.bin" % chunk), dtype=np.int32)
    assert p[0] == N and p[1] == L
    wb = p[3:3 + p[2]]
    return Chunk(seq, r, rpos, wb, bp, theta)


def synth_chunk(N, L, seed=1, block=100, jitter=True, budget=None, theta=0.001):
    """Synthetic block-coalescent chunk via the product library's generator."""
    from relate_amd import api
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L, dtype=np.float64)
    rpos = np.zeros(L + 1, dtype=np.float64)
    rc = lib.rl_synth_panel(N, L, C.c_uint64(seed), block, int(jitter),
                            seq.ctypes.data_as(C.c_void_p), None, 0,
                            bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                            rpos.ctypes.data_as(C.c_void_p))
    assert rc == 0
    if budget is None:
        wb = np.array([0, L], dtype=np.int32)
    else:
        wbuf = np.zeros(L + 2, dtype=np.int32)
        W = lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget),
                                 wbuf.ctypes.data_as(C.c_void_p), L)
        assert W > 0
        wb = wbuf[:W + 1].copy()
    return Chunk(seq, r, rpos, wb, bp, theta)


# ---------------------------------------------------------------- parsers
def parse_paint_file(path, N):
    """-> list over targets of dict(start,end,bb,la,alpha,be,lb,beta)"""
    buf = open(path, "rb").read()
    pos = 0
    out = []

    def stone(pos):
        isize, isub = struct.unpack_from("<QQ", buf, pos)
        assert isize == 1 and isub == N
        bsnp, ls, k = struct.unpack_from("<ifi", buf, pos + 16)
        u = np.frombuffer(buf, dtype=np.float32, count=k, offset=pos + 28)
        t = np.frombuffer(buf, dtype=np.int32, count=k, offset=pos + 28 + 4 * k)
        return bsnp, ls, np.repeat(u, t), pos + 28 + 8 * k

    for _ in range(N):
        start, end = struct.unpack_from("<ii", buf, pos)
        bb, la, alpha, p2 = stone(pos + 8)
        be, lb, beta, p3 = stone(p2)
        out.append(dict(start=start, end=end, bb=bb, la=la, alpha=alpha, be=be, lb=lb, beta=beta))
        pos = p3
    assert pos == len(buf)
    return out


def parse_anc(path):
    """-> (N, list of (pos, parents[2N-1], num_events, snp_begin, snp_end))"""
    buf = open(path, "rb").read()
    has_ages = buf[0]
    N, = struct.unpack_from("<I", buf, 1)
    pos = 5
    if has_ages:
        pos += 8 * N
    T, = struct.unpack_from("<I", buf, pos)
    pos += 4
    node = np.dtype([("parent", "<i4"), ("bl", "<f8"), ("ne", "<f4"), ("b", "<i4"), ("e", "<i4")])
    trees = []
    for _ in range(T):
        p, = struct.unpack_from("<i", buf, pos)
        pos += 4
        a = np.frombuffer(buf, dtype=node, count=2 * N - 1, offset=pos)
        pos += node.itemsize * (2 * N - 1)
        trees.append((p, a["parent"].copy(), a["ne"].copy(), a["b"].copy(), a["e"].copy()))
    assert pos == len(buf)
    return N, trees


def run_ref(args, cwd):
    """run the reference CLI built by `make -C oracle ref` (container only)"""
    return subprocess.run([REF_RELATE] + args, cwd=cwd, check=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def have_ref():
    return os.path.exists(REF_RELATE) and os.path.exists(REF_HARNESS)
