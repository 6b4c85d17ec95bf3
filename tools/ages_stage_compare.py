#!/usr/bin/env python3
"""tools/ages_stage_compare.py [N] [L]: Paint + BuildTopology of a synthetic chunk with a tenth of the samples ancient
(`--sample_ages`), once with the host tree builder and once with the device's (the AGES build of minmatch_gpu.hip):
wall-clock of the stage, trees per builder, and that the files are the same.  -> one JSON line (and
gpurun_out/ages_stage_compare.json)."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil  # noqa: E402
from bigtile import link_inputs  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2400
    cli = os.path.join(ROOT, "relate_amd", "Relate")
    ch = rlutil.synth_chunk(N, L, seed=41, budget=3e7 * (N / 1000.0) ** 2)
    rng = np.random.RandomState(9)
    ages = np.zeros(N)
    ages[rng.rand(N) < 0.1] = 800.0
    ages[rng.rand(N) < 0.04] = 2400.0
    ages = np.repeat(ages[::2], 2)
    out = {"N": N, "L": L, "sections": int(ch.W), "ancient_samples": int((ages > 0).sum())}
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "ages.txt"), "w") as f:
            f.write("\n".join("%g" % a for a in ages) + "\n")
        ch.write(os.path.join(tmp, "host", "out"))
        link_inputs(os.path.join(tmp, "host", "out"), os.path.join(tmp, "dev", "out"))
        for which, gpu in (("host", "0"), ("dev", "1")):
            env = dict(os.environ, RELATE_AMD_GPU_BUILD=gpu, RELATE_AMD_TIMING="1")
            t0 = time.time()
            p = subprocess.run([cli, "--mode", "PaintBuildTopology", "--chunk_index", "0", "--sample_ages", "../ages.txt",
                                "-o", "out"], cwd=os.path.join(tmp, which), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=env)
            dt = time.time() - t0
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            found = re.findall(r"(\d+) trees on the GPU, (\d+) on the host", p.stderr.decode())
            out[which] = {"seconds": round(dt, 2), "trees_on_device": sum(int(g) for g, _ in found),
                          "trees_on_host": sum(int(h) for _, h in found)}
        same = True
        for w in range(ch.W):
            for ext in ("anc", "mut"):
                a = open(os.path.join(tmp, "host", "out", "chunk_0", "out_%d.%s" % (w, ext)), "rb").read()
                b = open(os.path.join(tmp, "dev", "out", "chunk_0", "out_%d.%s" % (w, ext)), "rb").read()
                same = same and a == b
        out["files_identical"] = same
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ages_stage_compare.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
