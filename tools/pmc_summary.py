#!/usr/bin/env python3
"""HBM traffic (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/gpu_r06_i.sh)
of the Paint and RePaint kernels of `python3 bench.py --steps 1 --warmup 0 --skip-cpu --skip-chunk --skip-alt`.
rocprofv3 reports both counters in KB per dispatch (MI355X_MICROARCH.md, HBM section); the guide's gfx950 correction:
FETCH_SIZE under-reports loads that bypass the VGPRs (`global_load_lds_dwordx4`, 16 B per lane) by 2x -- applied to
the checkpoint rows the backward RePaint kernel reads that way (stated separately, `fetch_bytes_corrected`)."""
import glob, json, os, sqlite3, sys

base = sys.argv[1]


def db_of(sub):
    return sqlite3.connect(glob.glob(os.path.join(base, sub, "**", "*.db"), recursive=True)[0])


def per_kernel(sub, counter):
    out = {}
    for name, n, tot in db_of(sub).execute("select kernel_name, count(*), sum(value) from counters_collection "
                                           "where counter_name=? group by kernel_name", (counter,)):
        out[name] = (tot / n, n)
    return out


bench = json.loads(open(os.path.join(base, "bench_fetch.json")).read().strip().split("\n")[-1])
N = int(bench["config"]["workload"].split("N=")[1].split()[0])
L = int(bench["config"]["workload"].split("L=")[1].split()[0])
f, w = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
res = {"note": __doc__.strip().split("\n")[0], "N": N, "L": L,
       "algorithmic_read_bytes_per_paint_launch": 2.0 * N * bench["config"]["sum_k_D_k"] / 8.0, "kernels": {}}
for name in sorted(f):
    key = None
    if "paint_kernel<" in name and "repaint" not in name:
        args = name.split("<")[1].split(">")[0].split(",")
        mode = {"0": "lanes", "1": "exact", "2": "exact_serial"}[args[2].strip()]
        key = mode + {"0": "_fwd_alone", "1": "_bwd_alone", "2": ""}[args[4].strip()]
    elif "paint32_kernel<" in name:
        args = name.split("<")[1].split(">")[0].split(",")
        key = "lanes32" + {"0": "_fwd_alone", "1": "_bwd_alone", "2": ""}[args[3].strip()]
    elif "repaint_fwd_kernel" in name:
        key = "repaint_fwd"
    elif "repaint_bwd_kernel" in name:
        key = "repaint_bwd"
    if key is None:
        continue
    fb, wb = f[name][0] * 1024.0, (w.get(name) or (0, 0))[0] * 1024.0
    res["kernels"][key] = {"kernel": name[:100], "dispatches": f[name][1], "fetch_bytes": fb, "write_bytes": wb,
                           "hbm_bytes_per_launch": fb + wb}
    if key == "repaint_bwd":
        res["kernels"][key]["fetch_bytes_corrected"] = 2.0 * fb
        res["kernels"][key]["hbm_bytes_per_launch_corrected"] = 2.0 * fb + wb
k = res["kernels"]
if "repaint_fwd" in k and "repaint_bwd" in k:
    k["repaint"] = {"hbm_bytes_per_launch": k["repaint_fwd"]["hbm_bytes_per_launch"] + k["repaint_bwd"]["hbm_bytes_per_launch_corrected"],
                    "note": "forward + backward kernel of one window, backward fetch with the 2x correction"}
json.dump(res, open(os.path.join(base, "pmc_c3.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
