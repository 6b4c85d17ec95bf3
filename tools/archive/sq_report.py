"""per-step SQ counter report from tools/gpu_sq_r01.sh output (gpurun_out/sq)"""
import sqlite3, glob, json, sys
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sq"
for mode in ("exact", "lanes"):
    f = glob.glob(f"{base}/{mode}/**/*.db", recursive=True)[0]
    steps = json.load(open(f"{base}/{mode}_bench.json"))["config"]["sum_k_D_k"]
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    d = {}
    for k, c, v in rows:
        if "paint_kernel" in k:
            d.setdefault("bwd" if "true>" in k else "fwd", {})[c] = v / steps
    for k, v in sorted(d.items()):
        print(mode, k, " ".join("%s=%.0f" % (c.replace("SQ_", ""), x) for c, x in sorted(v.items())))
