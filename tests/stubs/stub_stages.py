"""Stand-in stages for relate_amd.dist (tests/test_dist_cpu.py::test_launcher_really_starts_two_ranks): the interface
of relate_amd.api's stage functions, each call recorded as a line `<rank> <stage> <chunk> [sections]` in
<out_dir>/calls_rank<rank>.txt -- no GPU, no library."""
import os


def _log(out_dir, what):
    rank = os.environ.get("RANK", "0")
    with open(os.path.join(out_dir, "calls_rank%s.txt" % rank), "a") as f:
        f.write("%s %s\n" % (rank, what))


def num_sections(out_dir, chunk_index=0):
    return 3 + chunk_index


def stage_paint(out_dir, chunk_index=0, painting=None, device=0, **kw):
    _log(out_dir, "paint %d dev%d" % (chunk_index, device))


def stage_build_topology(out_dir, chunk_index, first_section, last_section, painting=None, device=0, **kw):
    _log(out_dir, "build %d %d-%d dev%d" % (chunk_index, first_section, last_section, device))


def stage_paint_build_topology(out_dir, chunk_index, first_section, last_section, painting=None, device=0, **kw):
    _log(out_dir, "fused %d %d-%d dev%d" % (chunk_index, first_section, last_section, device))


def stage_find_equivalent_branches(out_dir, chunk_index=0):
    _log(out_dir, "feb %d" % chunk_index)


class Shard:
    """stand-in for relate_amd.api.Shard (`python -m relate_amd.dist --by-targets --stages stubs.stub_stages`): rows are a
    known function of (section, SNP, target, donor); an owner asks for a few SNPs per section and checks what comes back"""

    def __init__(self, out_dir, chunk_index, k_begin, k_end, **kw):
        import numpy as np
        self.np = np
        self.out_dir, self.chunk = out_dir, chunk_index
        self.N = int(np.fromfile(os.path.join(out_dir, "parameters_c%d.bin" % chunk_index), dtype=np.int32, count=1)[0])
        self.W = num_sections(out_dir, chunk_index)
        self.k_begin, self.k_end = k_begin, k_end

    def expected(self, section, snp):
        np = self.np
        k = np.arange(self.N, dtype=np.float32)[:, None]
        n = np.arange(self.N, dtype=np.float32)[None, :]
        return (1000.0 * section + snp + 0.25 * k + 0.001 * n).astype(np.float32)

    def set_window_rows(self, rows):
        pass

    def expect_builders(self, n):
        pass

    def rows(self, section, snp, ptr):
        import ctypes as C
        np = self.np
        out = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_float)), shape=(self.k_end - self.k_begin, self.N))
        out[:] = self.expected(section, snp)[self.k_begin:self.k_end]

    def release_section(self, section):
        _log(self.out_dir, "release %d %d" % (self.chunk, section))

    def copy_on_device(self, dst, src, nbytes):
        import ctypes as C
        C.memmove(dst, src, nbytes)

    def build_section(self, section, matrix, matrix_dev=None, build_device=None, no_consistency=False, fb=0):
        np = self.np
        d = np.zeros((self.N, self.N), dtype=np.float32)
        for snp in (10 * section, 10 * section + 3, 10 * section + 9):
            assert matrix(snp, d.ctypes.data) in (None, 0)
            assert np.array_equal(d, self.expected(section, snp)), (section, snp)
        _log(self.out_dir, "built %d %d" % (self.chunk, section))
        return 3

    def close(self):
        pass
