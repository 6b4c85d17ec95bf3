"""The C-ABI shared library loads and exports every symbol include/relate_amd.h
declares; the GPU entry points fail loudly (never fall back) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rlutil
from relate_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "relate_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported():
    lib = api.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "librelate_amd.so does not export %s" % s


def test_version_and_device_count():
    lib = api.lib()
    assert b"relate_amd" in lib.rl_version()
    assert api.device_count() >= 0


def test_no_cpu_fallback_without_gpu():
    if api.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(api.RelateError) as e:
        api.Context()
    assert "no usable HIP device" in str(e.value)
    rc = api.lib().rl_stage_paint(b"/nonexistent", 0, 0, 0.001, 1.0, 0, 0)
    assert rc != 0


def test_synth_panel_deterministic_and_chunk_files_roundtrip(tmp_path):
    a = rlutil.synth_chunk(40, 500, seed=9, budget=20000)
    b = rlutil.synth_chunk(40, 500, seed=9, budget=20000)
    assert np.array_equal(a.seq, b.seq) and np.array_equal(a.wb, b.wb) and a.W > 1
    f = (a.seq == ord("1")).mean()
    assert 0.05 < f < 0.4
    # the product's chunk-file writer and the test-side writer agree byte for byte
    lib = api.lib()
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir()
    a.write(str(d2))
    rc = lib.rl_write_chunk_files(str(d1).encode(), 0, a.N, a.L, a.seq.ctypes.data_as(C.c_void_p),
                                  a.bp.ctypes.data_as(C.c_void_p), a.r.ctypes.data_as(C.c_void_p),
                                  a.rpos.ctypes.data_as(C.c_void_p), a.wb.ctypes.data_as(C.c_void_p), a.W)
    assert rc == 0
    for fn in ["chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.state", "parameters_c0.bin"]:
        assert open(d1 / fn, "rb").read() == open(d2 / fn, "rb").read(), fn
    c = rlutil.read_chunk(str(d1))
    assert np.array_equal(c.seq, a.seq) and np.array_equal(c.wb, a.wb)


def test_stage_opts_that_never_saw_init_are_refused():
    """a zeroed rl_stage_opts (size 0) must not silently run with the defaults: RL_EINVAL before any
    device work, from all three *_ex stages"""
    lib = api.lib()
    raw = (C.c_ubyte * 256)()  # larger than the struct, all zero: size field 0
    for fn, args in ((lib.rl_stage_paint_ex, (b"/nonexistent", 0)),
                     (lib.rl_stage_build_topology_ex, (b"/nonexistent", 0, 0, 0)),
                     (lib.rl_stage_paint_build_topology_ex, (b"/nonexistent", 0, 0, 0))):
        fn.argtypes = [C.c_char_p] + [C.c_int] * (len(args) - 1) + [C.c_void_p]
        assert fn(*args, C.cast(raw, C.c_void_p)) == -1, fn  # RL_EINVAL
        assert b"rl_stage_opts_init" in lib.rl_last_error()
