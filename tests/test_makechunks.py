"""MakeChunks (host) against the reference binary, byte for byte: every chunk
file, parameter file and props.bin -- on the bundled example data and on a
synthetic .haps that is cut into several overlapping chunks.  Needs
oracle/_ref (build container); the committed example8 fixture carries the
single-chunk case to the GPU box (its chunk files were written by the
reference's MakeChunks)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import rlutil
from golden_util import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "relate_amd", "Relate")


def write_synth_haps(work, N, L, seed):
    ch = rlutil.synth_chunk(N, L, seed=seed, budget=None)
    with open(os.path.join(work, "s.haps"), "w") as f:
        for s in range(L):
            a, b = ("C", "T") if s % 3 else ("A", "C")
            f.write("1 rs%d %d %s %s %s\n" % (s, ch.bp[s], a, b, " ".join(chr(c) for c in ch.seq[s])))
    with open(os.path.join(work, "s.sample"), "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(N // 2):
            f.write("id%d id%d 0\n" % (i, i))
    with open(os.path.join(work, "s.map"), "w") as f:
        f.write("pos COMBINED_rate Genetic_Map\n")
        hi = int(ch.bp[-1]) + 200000
        for bp in range(0, hi, 40000):
            f.write("%d %.3f %.8f\n" % (bp, 1.0 + (bp // 40000) % 3, bp * 1.1e-6 + 0.003 * ((bp // 40000) % 5)))
    return ch


def compare_dirs(a, b):
    """a: the reference's directory, b: ours -- the same files, nothing else (the bit-packed panel chunk_<i>.bits is
    opt-in: test_bits_file_is_the_hap_file_bit_packed)"""
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    assert fa == fb
    for fn in fa:
        assert open(os.path.join(a, fn), "rb").read() == open(os.path.join(b, fn), "rb").read(), fn
    return fa


@pytest.mark.ref
@pytest.mark.skipif(not rlutil.have_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("N,L,memory,extra", [(6, 50000, "0.0005", []), (8, 3000, "0.0002", ["--transversion"]),
                                              (8, 3000, "0.00020001", []), (10, 4000, "0.0003333", [])])
def test_makechunks_matches_reference(tmp_path, N, L, memory, extra):
    work = str(tmp_path)
    write_synth_haps(work, N, L, seed=N)
    args = ["--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map", "--memory", memory]
    subprocess.run([rlutil.REF_RELATE] + args + extra + ["-o", "ref"], cwd=work, check=True, stderr=subprocess.PIPE)
    p = subprocess.run([CLI] + args + extra + ["-o", "ours"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    files = compare_dirs(os.path.join(work, "ref"), os.path.join(work, "ours"))
    assert "parameters.bin" in files and "props.bin" in files
    if L == 50000:
        assert "chunk_2.hap" in files  # several overlapping chunks
    # an existing output directory is refused, like the reference
    p = subprocess.run([CLI] + args + ["-o", "ours"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"already exists" in p.stderr


@pytest.mark.parametrize("tag,memory,extra", [("a", "0.0005", []), ("b", "0.0002", ["--transversion"]),
                                              ("c", "0.00020001", [])])
def test_makechunks_matches_committed_reference_outputs(tmp_path, tag, memory, extra):
    """runs anywhere: inputs regenerated from the seed (md5-checked), outputs against tests/golden/makechunks.npz --
    the md5 of every file the reference's MakeChunks wrote for them (tools/make_golden.py makechunks), the
    parameter files byte for byte"""
    import hashlib
    z = np.load(os.path.join(ROOT, "tests", "golden", "makechunks.npz"))
    N, L = [int(x) for x in z["%s/args" % tag]]
    work = str(tmp_path)
    write_synth_haps(work, N, L, seed=N)
    for fn in ("s.haps", "s.sample", "s.map"):
        assert hashlib.md5(open(os.path.join(work, fn), "rb").read()).digest() == z["%s/in_md5/%s" % (tag, fn)].tobytes(), fn
    p = subprocess.run([CLI, "--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map",
                        "--memory", memory] + extra + ["-o", "ours"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    want = sorted(k.split("/", 2)[2] for k in z.files if k.startswith(tag + "/md5/"))
    assert sorted(os.listdir(os.path.join(work, "ours"))) == want
    for fn in want:
        b = open(os.path.join(work, "ours", fn), "rb").read()
        if "%s/file/%s" % (tag, fn) in z.files:
            assert b == z["%s/file/%s" % (tag, fn)].tobytes(), fn
        assert hashlib.md5(b).digest() == z["%s/md5/%s" % (tag, fn)].tobytes(), fn
    if tag == "a":
        assert "chunk_2.hap" in want  # several overlapping chunks


@pytest.mark.skipif(not os.path.exists("/root/reference/example/data/example.haps.gz"),
                    reason="reference example data not present")
def test_makechunks_example_data_matches_fixture(tmp_path):
    # the example8 fixture's chunk files were produced by the reference's MakeChunks
    # from the first 3000 SNPs of example/data with a uniform 1 cM/Mb map (tools/make_golden.py)
    src = "/root/reference/example/data"
    work = str(tmp_path)
    lines = []
    with gzip.open(os.path.join(src, "example.haps.gz"), "rt") as f:
        for i, line in enumerate(f):
            if i >= 3000:
                break
            lines.append(line)
    open(os.path.join(work, "ex.haps"), "w").writelines(lines)
    subprocess.run("gunzip -c %s/example.sample.gz > %s/ex.sample" % (src, work), shell=True, check=True)
    first, last = int(lines[0].split()[2]), int(lines[-1].split()[2])
    with open(os.path.join(work, "ex.map"), "w") as f:
        f.write("pos COMBINED_rate Genetic_Map\n")
        for bp in range(max(0, first - 50000), last + 100000, 50000):
            f.write("%d 1.0 %.6f\n" % (bp, bp * 1e-6))
    p = subprocess.run([CLI, "--mode", "MakeChunks", "--haps", "ex.haps", "--sample", "ex.sample", "--map", "ex.map",
                        "--memory", "0.0002", "-o", "example"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    (tmp_path / "fx").mkdir()
    fx = Fixture("example8", tmp_path / "fx")
    for fn in ["parameters_c0.bin", "chunk_0.hap", "chunk_0.r", "chunk_0.rpos", "chunk_0.bp", "chunk_0.dist", "chunk_0.state"]:
        assert open(os.path.join(work, "example", fn), "rb").read() == fx.z["in/" + fn].tobytes(), fn


@pytest.mark.ref
@pytest.mark.skipif(not rlutil.have_ref() or not os.path.exists("/root/reference/example/data/example.haps.gz"),
                    reason="oracle/_ref or the reference example data not present")
def test_makechunks_gz_example_matches_reference(tmp_path):
    # the full bundled example (gzip input, N=8), cut into several chunks by a small --memory
    src = "/root/reference/example/data"
    work = str(tmp_path)
    with open(os.path.join(work, "ex.map"), "w") as f:
        f.write("pos COMBINED_rate Genetic_Map\n")
        for i, bp in enumerate(range(0, 60000000, 25000)):
            f.write("%d %.4f %.8f\n" % (bp, 0.5 + (i % 7) * 0.25, bp * 1.2e-6 + 1e-4 * (i % 11)))
    args = ["--mode", "MakeChunks", "--haps", src + "/example.haps.gz", "--sample", src + "/example.sample.gz",
            "--map", "ex.map", "--memory", "0.001"]
    subprocess.run([rlutil.REF_RELATE] + args + ["-o", "ref"], cwd=work, check=True, stderr=subprocess.PIPE)
    p = subprocess.run([CLI] + args + ["-o", "ours"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    files = compare_dirs(os.path.join(work, "ref"), os.path.join(work, "ours"))
    assert "chunk_1.hap" in files
    # an allowance too small for the 20000-SNP overlap aborts in the reference (data.cpp:170); here it is an error
    args[-1] = "0.0005"
    p = subprocess.run([CLI] + args + ["-o", "small"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stderr


def test_bits_file_is_the_hap_file_bit_packed(tmp_path):
    """chunk_<i>.bits (this library's side output of MakeChunks under RELATE_AMD_CHUNK_BITS=1, the layout
    rl_set_chunk_bits takes) against chunk_<i>.hap, the reference's char panel, for every chunk of a multi-chunk job
    (rows carried over the 20000-SNP overlap included).  The file names its .hap by size and modification time;
    without the option MakeChunks writes the reference's files only and REMOVES a .bits an earlier run left (a stale
    one would be read in place of the new .hap); FindEquivalentBranches removes it too (the reference's Finalize
    ends on an rmdir of the directory, Finalize.cpp:290)"""
    work = str(tmp_path)
    N, L = 70, 26000
    write_synth_haps(work, N, L, seed=3)
    args = ["--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map", "--memory", "0.0064"]
    p = subprocess.run([CLI] + args + ["-o", "ours"], cwd=work, stderr=subprocess.PIPE,
                       env=dict(os.environ, RELATE_AMD_CHUNK_BITS="1"))
    assert p.returncode == 0, p.stderr.decode()
    chunks = sorted(f for f in os.listdir(os.path.join(work, "ours")) if f.endswith(".hap"))
    assert len(chunks) >= 2
    for f in chunks:
        hap = open(os.path.join(work, "ours", f), "rb").read()
        hl, hn = np.frombuffer(hap, np.uint64, 2)
        seq = np.frombuffer(hap, np.uint8, int(hl * hn), 16).reshape(int(hl), int(hn))
        bits = open(os.path.join(work, "ours", f[:-4] + ".bits"), "rb").read()
        magic, bn, bl, rw = np.frombuffer(bits, np.uint32, 4)
        assert (magic, bn, bl, rw) == (0x32424c52, hn, hl, (hn + 31) // 32)
        st = os.stat(os.path.join(work, "ours", f))
        assert list(np.frombuffer(bits, np.uint64, 2, 16)) == [st.st_size, st.st_mtime_ns]
        words = np.frombuffer(bits, np.uint32, int(bl) * int(rw), 32).reshape(int(bl), int(rw))
        want = np.zeros((int(hl), int(rw) * 32), np.uint8)
        want[:, :int(hn)] = seq == ord("1")
        assert np.array_equal(np.packbits(want, axis=1, bitorder="little").view(np.uint32), words), f
    p = subprocess.run([CLI] + args + ["-o", "plain"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert not [f for f in os.listdir(os.path.join(work, "plain")) if f.endswith(".bits")]
    assert sorted(os.listdir(os.path.join(work, "plain"))) == sorted(
        f for f in os.listdir(os.path.join(work, "ours")) if not f.endswith(".bits"))
    # a second MakeChunks into a directory that holds .bits files, without the option: they go (rl_make_chunks writes
    # into an existing directory; the CLI stage refuses one)
    import ctypes as C
    from relate_amd import api
    lib = api.lib()
    lib.rl_make_chunks.argtypes = [C.c_char_p] * 5 + [C.c_int, C.c_float]
    env_before = os.environ.pop("RELATE_AMD_CHUNK_BITS", None)
    try:
        rc = lib.rl_make_chunks(os.path.join(work, "s.haps").encode(), os.path.join(work, "s.sample").encode(),
                                os.path.join(work, "s.map").encode(), None, os.path.join(work, "ours").encode(), 1,
                                C.c_float(0.0064))
    finally:
        if env_before is not None:
            os.environ["RELATE_AMD_CHUNK_BITS"] = env_before
    assert rc == 0, lib.rl_last_error()
    assert not [f for f in os.listdir(os.path.join(work, "ours")) if f.endswith(".bits")]


@pytest.mark.ref
@pytest.mark.skipif(not rlutil.have_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("variant", ["crlf", "tabs", "no_final_newline", "many_threads"])
def test_makechunks_line_formats_match_reference(tmp_path, variant):
    """the .haps text is taken apart by this library's own block reader (whole lines parsed on all host threads)
    where the reference uses fscanf + fgets: CR LF line ends, tabs and leading blanks, a last line without its
    newline (not a SNP for either), and more parser threads than lines in a block"""
    work = str(tmp_path)
    N, L = 8, 3000
    ch = write_synth_haps(work, N, L, seed=5)
    lines = open(os.path.join(work, "s.haps")).read().split("\n")[:-1]
    if variant == "crlf":
        text = "\r\n".join(lines) + "\r\n"
    elif variant == "tabs":
        text = "".join("  " + l.replace(" ", "\t") + "\n" for l in lines)
    elif variant == "no_final_newline":
        text = "\n".join(lines)
    else:
        text = "\n".join(lines) + "\n"
    with open(os.path.join(work, "s.haps"), "w", newline="") as f:
        f.write(text)
    args = ["--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map", "--memory", "0.0002"]
    subprocess.run([rlutil.REF_RELATE] + args + ["-o", "ref"], cwd=work, check=True, stderr=subprocess.PIPE)
    env = dict(os.environ, RELATE_AMD_THREADS="61") if variant == "many_threads" else dict(os.environ)
    p = subprocess.run([CLI] + args + ["-o", "ours"], cwd=work, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()
    compare_dirs(os.path.join(work, "ref"), os.path.join(work, "ours"))


@pytest.mark.parametrize("damage,message", [("short", b"alleles"), ("blank", b"malformed line 3"), ("letters", b"malformed line 3")])
def test_makechunks_reports_the_line_that_does_not_parse(tmp_path, damage, message):
    work = str(tmp_path)
    write_synth_haps(work, 8, 200, seed=6)
    lines = open(os.path.join(work, "s.haps")).read().split("\n")[:-1]
    if damage == "short":
        lines[2] = lines[2][:-2]          # one allele missing
    elif damage == "blank":
        lines[2] = ""
    else:
        f = lines[2].split(" ")
        f[2] = "x" + f[2]                 # the position is not a number
        lines[2] = " ".join(f)
    open(os.path.join(work, "s.haps"), "w").write("\n".join(lines) + "\n")
    p = subprocess.run([CLI, "--mode", "MakeChunks", "--haps", "s.haps", "--sample", "s.sample", "--map", "s.map",
                        "--memory", "0.0002", "-o", "ours"], cwd=work, stderr=subprocess.PIPE)
    assert p.returncode != 0 and message in p.stderr, p.stderr.decode()
