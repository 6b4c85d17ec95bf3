#!/usr/bin/env python3
"""HBM traffic of the Paint kernels from the two PMC passes of tools/gpu_profile_r01.sh
(gpurun_out/r01/fetch, gpurun_out/r01/write) -> profiles/r01_pmc_c3.json.
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB per dispatch (MI355X_MICROARCH.md, HBM section)."""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r01")


def per_kernel(sub, counter):
    db = sqlite3.connect(glob.glob(os.path.join(base, sub, "**", "*.db"), recursive=True)[0])
    out = {}
    for name, n, tot in db.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                   "where counter_name=? group by kernel_name", (counter,)):
        if "rl::paint_kernel<" in name:
            out[name] = tot / n * 1024.0
    return out


fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
bench = json.load(open(os.path.join(base, "bench_fetch.json")))
N = int(bench["config"]["workload"].split("N=")[1].split()[0])
L = int(bench["config"]["workload"].split("L=")[1].split()[0])
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_profile_r01.sh) of "
               "`python3 bench.py --steps 1 --warmup 0 --skip-cpu`; KB per dispatch * 1024, averaged over the "
               "dispatches of each kernel. Fetch is quoted raw (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide "
               "coalesced streams on gfx950; these kernels read the lane-mask rows by scalar loads and 4 B/lane "
               "touches). WRITE_SIZE matches the stepping-stone output W*N*N*4 B per direction. Fetch is far below "
               "the algorithmic bytes because a site's row is shared by every target derived there and is served "
               "by L2 / Infinity Cache.",
       "N": N, "L": L, "algorithmic_read_bytes_per_launch": N * bench["config"]["sum_k_D_k"] / 8.0, "kernels": {}}
for name in sorted(fetch):
    mode = {"0": "lanes", "1": "exact", "2": "exact_serial"}[name.split("<")[1].split(",")[2].strip()]
    key = mode + ("_bwd" if "true>" in name else "_fwd")
    res["kernels"][key] = {"kernel": name, "fetch_bytes": fetch[name], "write_bytes": write.get(name),
                           "hbm_bytes_per_launch": fetch[name] + (write.get(name) or 0.0)}
json.dump(res, open(os.path.join(ROOT, "profiles", "r01_pmc_c3.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
