"""SQ counters of the Paint launches per forward + backward step pair (one target at one visited site, both passes), from
tools/archive/gpu_r06_a.sh: the merged launch of each summation mode (the dispatches of the timed step AND of bench.py's
split-direction measurement are in the trace: the merged ones are the kernels with DIR = 2).
    python tools/sq_report.py <dir> <label>"""
import glob, json, sqlite3, sys
base, label = sys.argv[1], sys.argv[2]
print("# SQ counters per forward+backward step pair, merged Paint launch, %s; wave instructions / quad-cycles" % label)
for mode in ("exact", "lanes", "lanes32"):
    try:
        f = glob.glob(f"{base}/{mode}/**/*.db", recursive=True)[0]
        steps = json.loads(open(f"{base}/{mode}_bench.json").read().strip().split("\n")[-1])["config"]["sum_k_D_k"]
        db = sqlite3.connect(f)
        rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        for k in sorted(set(r[0] for r in rows)):
            if "paint" not in k or "repaint" in k:
                continue
            args = k.split("<")[1].split(">")[0].split(",")
            direction = args[-1].strip()
            vals = {c: v / n for kk, c, v, n in rows if kk == k}   # per dispatch
            lab = {"0": "forward alone", "1": "backward alone", "2": "merged (fwd+bwd)"}.get(direction, direction)
            print(mode, lab, k.split("(")[0].split("::")[-1][:40], " ".join("%s=%.0f" % (c.replace("SQ_", ""), x / steps) for c, x in sorted(vals.items())))
    except Exception as e:
        print(mode, "error:", e)
