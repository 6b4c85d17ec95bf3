"""Helpers shared by the big-tile tests against the reference binary (tests/test_n5000_gpu.py, tests/test_n10000_gpu.py):
regenerate the fixture's chunk from its seed (md5-checked against what the reference was given), run the drop-in CLI,
hold section 0's files to the reference's."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

import rlutil
from relate_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")


def md5(b):
    return np.frombuffer(hashlib.md5(b).digest(), dtype=np.uint8)


def make_chunk_dir(gold, work):
    """-> (fixture, W): <work>/out holds the chunk files the reference was given"""
    z = np.load(gold)
    N, L, W, seed = [int(x) for x in z["meta"]]
    mem = float(z["mem"][0])
    lib = api.lib()
    seq = np.zeros((L, N), dtype=np.uint8)
    bp = np.zeros(L, dtype=np.int32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    assert lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, seq.ctypes.data_as(C.c_void_p), None, 0,
                              bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                              rpos.ctypes.data_as(C.c_void_p)) == 0
    budget = mem * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wb = np.zeros(L + 2, dtype=np.int32)
    assert lib.rl_synth_windows(N, L, seq.ctypes.data_as(C.c_void_p), C.c_double(budget),
                                wb.ctypes.data_as(C.c_void_p), 499) == W
    assert np.array_equal(wb[:W + 1], z["wb"])
    d = os.path.join(work, "out")
    os.makedirs(d)
    lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    assert lib.rl_write_chunk_files(d.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p),
                                    bp.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                    rpos.ctypes.data_as(C.c_void_p), wb.ctypes.data_as(C.c_void_p), W) == 0
    for k in z.files:  # the chunk files are the ones the reference was given
        if k.startswith("in_md5/"):
            assert np.array_equal(md5(open(os.path.join(d, k[7:]), "rb").read()), z[k]), k
    return z, W


def link_inputs(src_out, dst_out):
    """a second output directory on the same chunk files"""
    os.makedirs(dst_out)
    for f in os.listdir(src_out):
        if os.path.isfile(os.path.join(src_out, f)):
            os.symlink(os.path.join(src_out, f), os.path.join(dst_out, f))


def check_section_0(z, out_dir, keep=False):
    anc = os.path.join(out_dir, "chunk_0", "out_0.anc")
    mut = open(os.path.join(out_dir, "chunk_0", "out_0.mut"), "rb").read()
    _, trees = rlutil.parse_anc(anc)
    assert [t[0] for t in trees] == list(z["tree_pos"]), "tree positions"
    for t, (tr, want) in enumerate(zip(trees, z["tree_parent_md5"])):
        if "tree_parent/%d" % t in z.files:
            assert np.array_equal(tr[1], z["tree_parent/%d" % t]), "parent array of tree %d" % t
        assert np.array_equal(md5(tr[1].astype("<i4").tobytes()), want), "parent array of tree %d" % t
    assert mut == z["mut/0"].tobytes()
    assert np.array_equal(md5(mut), z["md5/out_0.mut"])
    assert np.array_equal(md5(open(anc, "rb").read()), z["md5/out_0.anc"])
    if not keep:
        os.remove(anc)
        os.remove(os.path.join(out_dir, "chunk_0", "out_0.mut"))


def run_cli(mode, work, builder, extra_env=None, sections=(0, 0)):
    env = dict(os.environ)
    env["RELATE_AMD_GPU_BUILD"] = "1" if builder == "gpu" else "0"
    env["RELATE_AMD_TIMING"] = "1"
    env.update(extra_env or {})
    cmd = [CLI, "--mode", mode, "--chunk_index", "0", "-o", "out"]
    if mode != "Paint":
        cmd += ["--first_section", str(sections[0]), "--last_section", str(sections[1])]
    p = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stderr.decode()


def record(key, value):
    """what a test measured (not only asserted) goes to gpurun_out/test_reports.json -- or $RELATE_AMD_TEST_REPORT -- so
    that it survives `pytest -q`: tolerances used, identical-entry and identical-tree fractions of the fast modes
    (bench.py quotes profiles/r04_fast_modes.json, a committed copy)"""
    import json
    path = os.environ.get("RELATE_AMD_TEST_REPORT") or os.path.join(ROOT, "gpurun_out", "test_reports.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        data = json.load(open(path))
    except Exception:
        data = {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
