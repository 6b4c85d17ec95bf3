"""Every environment switch of the library that no other test file sets (README.md "Environment": fifteen switches +
the RELATE_AMD_TEST_* hooks), through the drop-in CLI on the reference-held fixture `synth70`: whatever the switch, the
.anc / .mut files are the reference's bytes."""
import os
import re
import subprocess
import sys

import pytest

from golden_util import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")


def build_topology(tmp_path, env, name="synth70", extra=()):
    work = tmp_path / "work"
    (work / "out").mkdir(parents=True)
    fx = Fixture(name, work / "out")
    fx.write_paint_files(str(work / "out" / "chunk_0" / "paint"))
    p = subprocess.run([CLI, "--mode", "BuildTopology", "--chunk_index", "0", "--first_section", "0", "--last_section",
                        str(fx.W - 1), "-o", "out"] + list(extra), cwd=str(work), stderr=subprocess.PIPE,
                       env=dict(os.environ, **env))
    assert p.returncode == 0, p.stderr.decode()[-800:]
    return fx, work, p.stderr.decode()


def same_as_reference(fx, work, anc_key="anc"):
    for w in range(fx.W):
        assert open(work / "out" / "chunk_0" / ("out_%d.mut" % w), "rb").read() == fx.z["mut/%d" % w].tobytes(), w
        assert open(work / "out" / "chunk_0" / ("out_%d.anc" % w), "rb").read() == fx.z["%s/%d" % (anc_key, w)].tobytes(), w


def test_lib(tmp_path):
    """RELATE_AMD_LIB: relate_amd.api loads the library it names (a copy of the in-tree build; a wrong path fails loudly)"""
    import shutil
    lib = tmp_path / "librelate_copy.so"
    shutil.copy(os.path.join(ROOT, "relate_amd", "librelate_amd.so"), lib)
    code = "import sys; sys.path.insert(0, %r); from relate_amd import api; api.lib(); print(api.lib()._name)" % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_LIB=str(lib)))
    assert p.returncode == 0 and str(lib) in p.stdout.decode(), p.stderr.decode()[-400:]
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_LIB=str(tmp_path / "not_there.so")))
    assert p.returncode != 0


@pytest.mark.gpu
def test_timing_levels(tmp_path):
    """RELATE_AMD_TIMING=1: the stage's phase lines; =2: the tree builder's progress marks as well; unset: neither"""
    fx, work, err = build_topology(tmp_path / "a", {"RELATE_AMD_GPU_BUILD": "1", "RELATE_AMD_TIMING": "2"})
    same_as_reference(fx, work)
    assert "[stage] sections" in err and "[mm trace]" in err
    fx, work, err = build_topology(tmp_path / "b", {"RELATE_AMD_GPU_BUILD": "1", "RELATE_AMD_TIMING": "1"})
    assert "[stage] sections" in err and "[mm trace]" not in err
    assert re.search(r"\((\d+) workers asked for\)", err)
    env = {k: v for k, v in os.environ.items() if k != "RELATE_AMD_TIMING"}
    fx, work, err = build_topology(tmp_path / "c", {"RELATE_AMD_GPU_BUILD": "1"})
    if "RELATE_AMD_TIMING" not in os.environ:
        assert "[stage]" not in err


@pytest.mark.gpu
def test_threads(tmp_path):
    """RELATE_AMD_THREADS: the host threads of the plan and the encoders (one, and more than the fixture has targets)"""
    for n in ("1", "61"):
        work = tmp_path / n
        (work / "out").mkdir(parents=True)
        fx = Fixture("synth70", work / "out")
        p = subprocess.run([CLI, "--mode", "Paint", "--chunk_index", "0", "-o", "out"], cwd=str(work), stderr=subprocess.PIPE,
                           env=dict(os.environ, RELATE_AMD_THREADS=n))
        assert p.returncode == 0, p.stderr.decode()[-400:]
        for w in range(fx.W):
            assert open(work / "out" / "chunk_0" / "paint" / ("relate_%d.bin" % w), "rb").read() == fx.paint_file(w), w


@pytest.mark.gpu
@pytest.mark.parametrize("occ", ["1", "2"])
def test_build_workers_and_occ(tmp_path, occ):
    """RELATE_AMD_BUILD_WORKERS / RELATE_AMD_BUILD_OCC: fewer resident workers than sections, one or two workgroups of
    the worker kernel per CU -- the same trees"""
    fx, work, err = build_topology(tmp_path, {"RELATE_AMD_GPU_BUILD": "1", "RELATE_AMD_BUILD_WORKERS": "2",
                                              "RELATE_AMD_BUILD_OCC": occ, "RELATE_AMD_TIMING": "1"})
    same_as_reference(fx, work)
    goals = [int(x) for x in re.findall(r"goal (\d+)", err)]
    assert goals and max(goals) <= 2, goals


@pytest.mark.gpu
def test_window_parts(tmp_path):
    """RELATE_AMD_WINDOW_PARTS: the smallest share of its rows a window may keep when the stage sizes the windows
    itself (here nothing forces bounded windows: the switch must not change a byte)"""
    fx, work, _ = build_topology(tmp_path, {"RELATE_AMD_GPU_BUILD": "1", "RELATE_AMD_WINDOW_PARTS": "4"})
    same_as_reference(fx, work)


@pytest.mark.gpu
def test_pin(tmp_path):
    """RELATE_AMD_PIN=0: the host builder's threads are left to the scheduler"""
    fx, work, _ = build_topology(tmp_path, {"RELATE_AMD_GPU_BUILD": "0", "RELATE_AMD_PIN": "0", "RELATE_AMD_BUILD_THREADS": "3"})
    same_as_reference(fx, work)


@pytest.mark.gpu
def test_fused_feb(tmp_path):
    """RELATE_AMD_FUSED_FEB=1: FindEquivalentBranches behind BuildTopology in the same call -- the .anc files as that
    stage leaves them"""
    fx, work, _ = build_topology(tmp_path, {"RELATE_AMD_FUSED_FEB": "1"})
    same_as_reference(fx, work, anc_key="feb_anc")


@pytest.mark.gpu
def test_handover_hook(tmp_path):
    """RELATE_AMD_TEST_HANDOVER_EVERY=3: every third device build is handed to the host builder with the carried state,
    as a tree with more tied candidates than the device's lists hold is"""
    fx, work, err = build_topology(tmp_path, {"RELATE_AMD_GPU_BUILD": "1", "RELATE_AMD_TEST_HANDOVER_EVERY": "3",
                                              "RELATE_AMD_TIMING": "1"})
    same_as_reference(fx, work)
    on_host = sum(int(x) for x in re.findall(r"trees on the GPU, (\d+) on the host", err))
    assert on_host > 0
