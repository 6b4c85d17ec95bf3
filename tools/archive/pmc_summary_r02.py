#!/usr/bin/env python3
"""Round-2 evidence -> profiles/: HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ counters of the Paint launch and
of K2 from the rocprofv3 runs of tools/gpu_profile_r02.sh (gpurun_out/r02).  rocprofv3 reports FETCH_SIZE / WRITE_SIZE
in KB per dispatch (MI355X_MICROARCH.md, HBM section)."""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02")


def db_of(sub):
    return sqlite3.connect(glob.glob(os.path.join(base, sub, "**", "*.db"), recursive=True)[0])


def per_kernel(sub, counter, pick):
    out = {}
    for name, n, tot in db_of(sub).execute("select kernel_name, count(*), sum(value) from counters_collection "
                                           "where counter_name=? group by kernel_name", (counter,)):
        if pick in name:
            out[name] = (tot / n, n)
    return out


bench = json.load(open(os.path.join(base, "bench_fetch.json")))
N = int(bench["config"]["workload"].split("N=")[1].split()[0])
L = int(bench["config"]["workload"].split("L=")[1].split()[0])
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_profile_r02.sh) of `python3 "
               "bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt`; KB per dispatch * 1024, averaged "
               "over the dispatches of the kernel (the bench's split-launch measurement adds single-direction "
               "dispatches of the same kernel template with DIR = 0 / 1; the merged launch is DIR = 2).",
       "N": N, "L": L, "algorithmic_read_bytes_per_launch": 2.0 * N * bench["config"]["sum_k_D_k"] / 8.0,
       "kernels": {}}
for pick, key in (("rl::paint_kernel<", "paint"), ("rl::repaint_kernel<", "repaint")):
    f, w = per_kernel("fetch", "FETCH_SIZE", pick), per_kernel("write", "WRITE_SIZE", pick)
    for name in sorted(f):
        k = key
        if key == "paint":
            args = name.split("<")[1].split(">")[0].split(",")
            mode = {"0": "lanes", "1": "exact", "2": "exact_serial"}[args[2].strip()]
            dirn = {"0": "_fwd_alone", "1": "_bwd_alone", "2": ""}[args[4].strip()]
            k = mode + dirn
        res["kernels"][k] = {"kernel": name, "dispatches": f[name][1], "fetch_bytes": f[name][0] * 1024.0,
                             "write_bytes": (w.get(name) or (0, 0))[0] * 1024.0,
                             "hbm_bytes_per_launch": (f[name][0] + (w.get(name) or (0, 0))[0]) * 1024.0}
json.dump(res, open(os.path.join(ROOT, "profiles", "r02_pmc_c3.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
# SQ counters of the Paint launch, per (target, site) step and wavefront
try:
    sq = json.load(open(os.path.join(base, "bench_sq.json")))
    steps = sq["config"]["sum_k_D_k"]
    rows = db_of("sq").execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                               "group by kernel_name, counter_name").fetchall()
    lines = ["# SQ counters of the Paint launch on an L=%d cut of the C3 panel (tools/gpu_profile_r02.sh), per step of a "
             "wavefront (a launch makes 2 * sum_k D_k = %d steps: forward + backward)" % (
                 int(sq["config"]["workload"].split("L=")[1].split()[0]), 2 * steps)]
    for k, c, v, n in sorted(rows):
        if "paint_kernel" in k:
            args = k.split("<")[1].split(">")[0].split(",")
            per = v / n / (steps * (2 if args[4].strip() == "2" else 1))
            lines.append("%-60s %-22s %12.1f per step (%d dispatches)" % (k[:60], c, per, n))
    open(os.path.join(ROOT, "profiles", "r02_sq_paint.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
except Exception as e:
    print("no SQ pass:", e)
