"""ctypes binding of librelate_amd.so (include/relate_amd.h).

This is the Python-side mirror of the C ABI: thin wrappers, numpy in / numpy
out.  There is no CPU fallback: if the shared library is missing, or no GPU is
visible when a GPU entry point is called, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librelate_amd.so")

RL_SUM_EXACT = 0
RL_SUM_LANES = 1
RL_SUM_EXACT_SERIAL = 2
RL_SUM_LANES32 = 3

_lib = None


class RelateError(RuntimeError):
    pass


def lib():
    """the loaded shared library (raises if it has not been built)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RelateError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C relate_amd/csrc)" % LIB_PATH)
        # (RELATE_AMD_LIB: another build of the library, tools/bench_builder_variants.py)
        L = C.CDLL(os.environ.get("RELATE_AMD_LIB") or LIB_PATH)
        L.rl_last_error.restype = C.c_char_p
        L.rl_version.restype = C.c_char_p
        L.rl_create.restype = C.c_void_p
        L.rl_create.argtypes = [C.c_int]
        L.rl_destroy.argtypes = [C.c_void_p]
        L.rl_total_sites.restype = C.c_longlong
        L.rl_total_sites.argtypes = [C.c_void_p]
        L.rl_window_open.restype = C.c_void_p
        L.rl_window_open.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.rl_window_open_bounded.restype = C.c_void_p
        L.rl_window_open_bounded.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_longlong,
                                             C.c_void_p]
        L.rl_window_close.argtypes = [C.c_void_p]
        L.rl_stage_paint.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.rl_stage_build_topology.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                              C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rl_quickbuild.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
        L.rl_set_painting.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.rl_window_matrix_rows_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RelateError("librelate_amd error %d: %s" % (rc, lib().rl_last_error().decode()))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().rl_device_count()


class Context:
    """One chunk on one GPU (rl_ctx)."""

    def __init__(self, device=0):
        self._h = lib().rl_create(device)
        if not self._h:
            raise RelateError(lib().rl_last_error().decode())
        self.N = self.L = self.W = 0

    def close(self):
        if self._h:
            lib().rl_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _dims(self):
        n, l, w = C.c_int(), C.c_int(), C.c_int()
        _check(lib().rl_chunk_dims(C.c_void_p(self._h), C.byref(n), C.byref(l), C.byref(w)))
        self.N, self.L, self.W = n.value, l.value, w.value

    def load_chunk(self, out_dir, chunk_index=0):
        _check(lib().rl_load_chunk(C.c_void_p(self._h), out_dir.encode(), chunk_index))
        self._dims()

    def set_chunk(self, seq, r, rpos, wb):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        L, N = seq.shape
        r = np.ascontiguousarray(r, dtype=np.float64)
        rpos = np.ascontiguousarray(rpos, dtype=np.float64)
        wb = np.ascontiguousarray(wb, dtype=np.int32)
        assert len(r) == L and len(rpos) == L + 1
        _check(lib().rl_set_chunk(C.c_void_p(self._h), N, L, _p(seq), _p(r), _p(rpos), _p(wb), len(wb) - 1))
        self._dims()

    def set_chunk_bits(self, N, bits, r, rpos, wb):
        bits = np.ascontiguousarray(bits, dtype=np.uint32)
        L, rw = bits.shape
        r = np.ascontiguousarray(r, dtype=np.float64)
        rpos = np.ascontiguousarray(rpos, dtype=np.float64)
        wb = np.ascontiguousarray(wb, dtype=np.int32)
        _check(lib().rl_set_chunk_bits(C.c_void_p(self._h), N, L, _p(bits), rw, _p(r), _p(rpos), _p(wb),
                                       len(wb) - 1))
        self._dims()

    def set_painting(self, theta, rho):
        _check(lib().rl_set_painting(C.c_void_p(self._h), theta, rho))

    def set_target_range(self, k_begin, k_end):
        """shard ONE chunk by target haplotype: this context handles targets k_begin .. k_end-1"""
        _check(lib().rl_set_target_range(C.c_void_p(self._h), int(k_begin), int(k_end)))

    def target_range(self):
        a, b = C.c_int(), C.c_int()
        _check(lib().rl_target_range(C.c_void_p(self._h), C.byref(a), C.byref(b)))
        return a.value, b.value

    def total_sites(self):
        v = lib().rl_total_sites(C.c_void_p(self._h))
        if v < 0:
            _check(int(v))
        return v

    def prepare(self):
        """plan + uploads + allocations, so that paint() times only the kernels"""
        _check(lib().rl_prepare(C.c_void_p(self._h)))

    def paint(self, sum_mode=RL_SUM_EXACT):
        """-> kernel milliseconds"""
        ms = C.c_float(0)
        _check(lib().rl_paint(C.c_void_p(self._h), sum_mode, C.byref(ms)))
        return ms.value

    def set_paint_split(self, split):
        """one launch per direction (so that paint_times() has something to report) instead of one for both"""
        _check(lib().rl_set_paint_split(C.c_void_p(self._h), int(split)))

    @property
    def tile(self):
        s, w = C.c_int(), C.c_int()
        _check(lib().rl_register_tile(C.c_void_p(self._h), C.byref(s), C.byref(w)))
        return s.value

    @property
    def waves(self):
        s, w = C.c_int(), C.c_int()
        _check(lib().rl_register_tile(C.c_void_p(self._h), C.byref(s), C.byref(w)))
        return w.value

    def paint_times(self):
        """-> (forward kernel ms, backward kernel ms) of the last paint()"""
        f, b = C.c_float(0), C.c_float(0)
        _check(lib().rl_paint_times(C.c_void_p(self._h), C.byref(f), C.byref(b)))
        return f.value, b.value

    def stones(self, w):
        """stepping stones of window w: one row per target of the context (all N by default)"""
        N = self.N
        k0, k1 = self.target_range()
        n = k1 - k0
        out = dict(alpha=np.empty((n, N), np.float32), beta=np.empty((n, N), np.float32),
                   ls_alpha=np.empty(n, np.float32), ls_beta=np.empty(n, np.float32),
                   bsnp_begin=np.empty(n, np.int32), bsnp_end=np.empty(n, np.int32))
        _check(lib().rl_get_stones(C.c_void_p(self._h), w, _p(out["alpha"]), _p(out["beta"]),
                                   _p(out["ls_alpha"]), _p(out["ls_beta"]), _p(out["bsnp_begin"]),
                                   _p(out["bsnp_end"])))
        return out

    def write_paint_files(self, paint_dir):
        os.makedirs(paint_dir, exist_ok=True)
        _check(lib().rl_write_paint_files(C.c_void_p(self._h), paint_dir.encode()))

    def write_paint_file(self, w, path):
        """window w's paint file alone (rl_write_paint_file)"""
        _check(lib().rl_write_paint_file(C.c_void_p(self._h), int(w), path.encode()))

    def paint_record(self, w, k):
        """-> bytes: target k's record of window w's paint file (rl_paint_record)"""
        n = C.c_size_t(0)
        _check(lib().rl_paint_record(C.c_void_p(self._h), int(w), int(k), None, C.c_size_t(0), C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(lib().rl_paint_record(C.c_void_p(self._h), int(w), int(k), buf, C.c_size_t(n.value), C.byref(n)))
        return buf.raw[:n.value]

    def open_window(self, w, paint_file=None, first_snp=None, sum_mode=RL_SUM_EXACT, max_rows=0):
        """max_rows > 0: keep at most that many posterior rows resident (rl_window_open_bounded)"""
        return Window(self, w, paint_file, first_snp, sum_mode, max_rows)


class Window:
    """DistanceMeasure for one window (rl_window): topology resident in HBM."""

    def __init__(self, ctx, w, paint_file, first_snp, sum_mode, max_rows=0):
        self.ctx = ctx
        ms = C.c_float(0)
        fs = -1 if first_snp is None else int(first_snp)
        self._h = lib().rl_window_open_bounded(C.c_void_p(ctx._h), w, paint_file.encode() if paint_file else None,
                                               fs, sum_mode, int(max_rows), C.byref(ms))
        if not self._h:
            raise RelateError(lib().rl_last_error().decode())
        self.repaint_ms = ms.value
        s, e = C.c_int(), C.c_int()
        _check(lib().rl_window_bounds(C.c_void_p(self._h), C.byref(s), C.byref(e)))
        self.start, self.end = s.value, e.value

    def close(self):
        if self._h:
            lib().rl_window_close(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        self.close()

    def rows(self, n):
        return lib().rl_window_rows(C.c_void_p(self._h), n)

    @property
    def repaints(self):
        """times RePaintSection has run for this window (> 1 only for a bounded one)"""
        return lib().rl_window_repaints(C.c_void_p(self._h))

    def topology(self, n):
        D, N = self.rows(n), self.ctx.N
        top = np.empty((D, N), np.float32)
        ls = np.empty(D, np.float32)
        _check(lib().rl_window_get_topology(C.c_void_p(self._h), n, _p(top), _p(ls)))
        return top, ls

    def advance(self, snp):
        _check(lib().rl_window_advance(C.c_void_p(self._h), snp))

    def matrix(self, snp):
        """distance matrix at snp: the rows of the context's targets (the whole N x N matrix by default)"""
        N = self.ctx.N
        k0, k1 = self.ctx.target_range()
        d = np.empty((k1 - k0, N), np.float32)
        ms = C.c_float(0)
        _check(lib().rl_window_matrix(C.c_void_p(self._h), snp, _p(d), C.byref(ms)))
        self.matrix_ms = ms.value
        return d

    def matrix_rows_into(self, snp, device_ptr):
        """the same rows written to a device buffer ((k_end-k_begin)*N floats), e.g. a torch tensor's
        data_ptr(): the send buffer of relate_amd.dist.all_gather_rows"""
        ms = C.c_float(0)
        _check(lib().rl_window_matrix_rows_device(C.c_void_p(self._h), snp, C.c_void_p(device_ptr), C.byref(ms)))
        self.matrix_ms = ms.value


def quickbuild(d, theta=0.001, prior=None):
    """MinMatch::QuickBuild on an N x N float matrix -> parent array (2N-1)"""
    d = np.array(d, dtype=np.float32, order="C")
    N = d.shape[0]
    parent = np.empty(2 * N - 1, np.int32)
    pr = None if prior is None else np.ascontiguousarray(prior, dtype=np.float32)
    _check(lib().rl_quickbuild(N, theta, _p(d), _p(pr), _p(parent), None, None))
    return parent


class Builder:
    """One MinMatch for a sequence of trees (rl_builder): device=None builds on the host, an int on that GPU."""

    def __init__(self, N, theta=0.001, device=None):
        L = lib()
        L.rl_builder_create.restype = C.c_void_p
        L.rl_builder_create.argtypes = [C.c_int, C.c_double, C.c_int]
        L.rl_builder_build.argtypes = [C.c_void_p] * 6
        L.rl_builder_last_on_gpu.argtypes = [C.c_void_p]
        L.rl_builder_destroy.argtypes = [C.c_void_p]
        self.N = N
        self._h = L.rl_builder_create(N, theta, -1 if device is None else int(device))
        if not self._h:
            raise RelateError(L.rl_last_error().decode())

    def build(self, d, prior=None):
        """-> (parent[2N-1], child_left[N-1], child_right[N-1])"""
        N = self.N
        d = np.array(d, dtype=np.float32, order="C")
        pr = None if prior is None else np.ascontiguousarray(prior, dtype=np.float32)
        parent = np.empty(2 * N - 1, np.int32)
        cl = np.empty(N - 1, np.int32)
        cr = np.empty(N - 1, np.int32)
        _check(lib().rl_builder_build(C.c_void_p(self._h), _p(d), _p(pr), _p(parent), _p(cl), _p(cr)))
        return parent, cl, cr

    def set_sample_ages(self, ages):
        ages = np.ascontiguousarray(ages, dtype=np.float64)
        _check(lib().rl_builder_set_sample_ages(C.c_void_p(self._h), _p(ages), len(ages)))

    @property
    def last_on_gpu(self):
        return lib().rl_builder_last_on_gpu(C.c_void_p(self._h)) == 1

    def close(self):
        if self._h:
            lib().rl_builder_destroy(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        self.close()


MATRIX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)  # rl_matrix_fn / rl_matrix_dev_fn


class Shard:
    """One rank's share of a chunk sharded by target haplotype (rl_shard, BASELINE.json config #5): the chunk loaded
    for targets [k_begin, k_end), their stepping stones painted and resident (from_paint_files=False) or read from
    the Paint stage's files.  rows(): this shard's rows of a section's distance matrix, to a device buffer;
    build_section(): the tree-sequence loop of a section this rank owns, its matrices supplied by callbacks."""

    def __init__(self, out_dir, chunk_index, k_begin, k_end, painting=None, sum_mode=RL_SUM_EXACT, device=0,
                 from_paint_files=False):
        L = lib()
        L.rl_shard_open.restype = C.c_void_p
        L.rl_shard_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                    C.c_int, C.c_int]
        L.rl_shard_close.argtypes = [C.c_void_p]
        L.rl_shard_dims.argtypes = [C.c_void_p] * 6
        L.rl_shard_section_bounds.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rl_shard_set_window_rows.argtypes = [C.c_void_p, C.c_longlong]
        L.rl_shard_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rl_shard_release_section.argtypes = [C.c_void_p, C.c_int]
        L.rl_shard_expect_builders.argtypes = [C.c_void_p, C.c_int]
        L.rl_shard_build_section.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, MATRIX_FN, MATRIX_FN,
                                             C.c_void_p, C.c_void_p]
        L.rl_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        th, rho = painting if painting else (0.001, 1.0)
        self.device = device
        self._h = L.rl_shard_open(out_dir.encode(), chunk_index, k_begin, k_end, 1 if painting else 0, th, rho,
                                  sum_mode, device, 1 if from_paint_files else 0)
        if not self._h:
            raise RelateError(L.rl_last_error().decode())
        v = [C.c_int() for _ in range(5)]
        _check(L.rl_shard_dims(C.c_void_p(self._h), *[C.byref(x) for x in v]))
        self.N, self.L, self.W, self.k_begin, self.k_end = [x.value for x in v]

    def close(self):
        if self._h:
            lib().rl_shard_close(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        self.close()

    def section_bounds(self, section):
        a, b = C.c_int(), C.c_int()
        _check(lib().rl_shard_section_bounds(C.c_void_p(self._h), section, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_window_rows(self, rows):
        _check(lib().rl_shard_set_window_rows(C.c_void_p(self._h), int(rows)))

    def rows(self, section, snp, device_ptr):
        """rows k_begin..k_end-1 of the section's distance matrix at snp -> (k_end-k_begin)*N floats at device_ptr"""
        _check(lib().rl_shard_rows(C.c_void_p(self._h), section, snp, C.c_void_p(device_ptr)))

    def release_section(self, section):
        _check(lib().rl_shard_release_section(C.c_void_p(self._h), section))

    def expect_builders(self, n):
        _check(lib().rl_shard_expect_builders(C.c_void_p(self._h), int(n)))

    def copy_on_device(self, dst_ptr, src_ptr, nbytes):
        _check(lib().rl_device_copy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes, self.device))

    def build_section(self, section, matrix, matrix_dev=None, build_device=None, no_consistency=False, fb=0):
        """matrix(snp, host_ptr) / matrix_dev(snp, device_ptr): fill the N x N float matrix of `snp` (return None or 0;
        an exception fails the section).  -> number of trees.  Blocks; releases the GIL while the trees are built."""
        errors = []

        def wrap(fn):
            def cb(_user, snp, ptr):
                try:
                    return int(fn(snp, ptr) or 0)
                except BaseException as e:  # (must not propagate through the C frames)
                    errors.append(e)
                    return -1
            return MATRIX_FN(cb)

        c_host = wrap(matrix)
        c_dev = wrap(matrix_dev) if matrix_dev is not None else C.cast(None, MATRIX_FN)
        n = C.c_int(0)
        rc = lib().rl_shard_build_section(C.c_void_p(self._h), section, 1 if no_consistency else 0, fb,
                                          -1 if build_device is None else int(build_device), c_host, c_dev, None,
                                          C.byref(n))
        if errors:
            raise errors[0]
        _check(rc)
        return n.value


def stage_paint(out_dir, chunk_index=0, painting=None, sum_mode=RL_SUM_EXACT, device=0):
    th, rho = painting if painting else (0.001, 1.0)
    _check(lib().rl_stage_paint(out_dir.encode(), chunk_index, 1 if painting else 0, th, rho, sum_mode, device))


def stage_build_topology(out_dir, chunk_index, first_section, last_section, painting=None, no_consistency=False,
                         fb=0, sum_mode=RL_SUM_EXACT, device=0):
    th, rho = painting if painting else (0.001, 1.0)
    _check(lib().rl_stage_build_topology(out_dir.encode(), chunk_index, first_section, last_section,
                                         1 if painting else 0, th, rho, 1 if no_consistency else 0, fb,
                                         sum_mode, device))


# this module's stage_paint_build_topology takes find_equivalent_branches=True (relate_amd.dist.run_chunks asks)
FUSED_FEB = True


def stage_paint_build_topology(out_dir, chunk_index, first_section, last_section, painting=None,
                               no_consistency=False, fb=0, sum_mode=RL_SUM_EXACT, device=0,
                               find_equivalent_branches=False):
    """Paint + BuildTopology of a chunk with the stepping stones kept in HBM (no paint files);
    find_equivalent_branches: the stage downstream fused in (every .anc written once, as that stage leaves it)"""
    if find_equivalent_branches:
        o = stage_opts(painting=painting, flags=1 if no_consistency else 0, fb=fb, sum_mode=sum_mode, device=device,
                       find_equivalent_branches=1)
        return stage_build_topology_ex(out_dir, chunk_index, first_section, last_section, o, fused=True)
    th, rho = painting if painting else (0.001, 1.0)
    f = lib().rl_stage_paint_build_topology
    f.argtypes = lib().rl_stage_build_topology.argtypes
    _check(f(out_dir.encode(), chunk_index, first_section, last_section, 1 if painting else 0, th, rho,
             1 if no_consistency else 0, fb, sum_mode, device))


class StageOpts(C.Structure):
    """rl_stage_opts (include/relate_amd.h): every option of a stage call, per call"""
    _fields_ = [("size", C.c_size_t), ("sum_mode", C.c_int), ("device", C.c_int), ("use_painting", C.c_int),
                ("theta", C.c_double), ("rho", C.c_double), ("flags", C.c_int), ("fb", C.c_int),
                ("sample_ages_path", C.c_char_p), ("gpu_build", C.c_int), ("window_rows", C.c_longlong),
                ("window_parts", C.c_int), ("section_threads", C.c_int), ("workers", C.c_int),
                ("repaint_lanes", C.c_int), ("park_stones", C.c_int), ("pin_threads", C.c_int),
                ("find_equivalent_branches", C.c_int)]


def stage_opts(**kw):
    """-> StageOpts with rl_stage_opts_init's defaults and the given fields (painting=(theta, rho) sets three)"""
    o = StageOpts()
    lib().rl_stage_opts_init.argtypes = [C.c_void_p]
    lib().rl_stage_opts_init(C.byref(o))
    assert o.size == C.sizeof(StageOpts), "StageOpts is out of step with include/relate_amd.h"
    painting = kw.pop("painting", None)
    if painting:
        o.use_painting, o.theta, o.rho = 1, painting[0], painting[1]
    for k, v in kw.items():
        if k == "sample_ages_path" and v is not None:
            v = v.encode()
        setattr(o, k, v)
    return o


def stage_build_topology_ex(out_dir, chunk_index, first_section, last_section, opts, fused=False):
    """rl_stage_build_topology_ex / rl_stage_paint_build_topology_ex (fused=True) with a StageOpts"""
    f = lib().rl_stage_paint_build_topology_ex if fused else lib().rl_stage_build_topology_ex
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _check(f(out_dir.encode(), chunk_index, first_section, last_section, C.byref(opts)))


def stage_find_equivalent_branches(out_dir, chunk_index=0):
    _check(lib().rl_stage_find_equivalent_branches(out_dir.encode(), chunk_index))


def num_sections(out_dir, chunk_index=0):
    """number of windows (= BuildTopology sections) of a chunk, from parameters_c<chunk>.bin"""
    p = np.fromfile(os.path.join(out_dir, "parameters_c%d.bin" % chunk_index), dtype=np.int32, count=3)
    return int(p[2]) - 1
