"""experiment: 40 sections of the N=5000 x L=100000 chunk; where the section threads' time goes (sums over sections)"""
import json, re, subprocess, sys, os
env = dict(os.environ)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    env[k] = v
p = subprocess.run([sys.executable, "tools/chunk_wallclock_big.py", "5000", "100000", "20", "40"], stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE, env=env)
d = json.loads(p.stdout.decode().strip().split("\n")[-1])
acc = {}
for line in d.get("build_topology_phases", []):
    for key, pat in (("matrices", r"distance matrices ([0-9.]+) s"), ("prior", r"clade prior ([0-9.]+) s"),
                     ("minmatch", r"MinMatch ([0-9.]+) s"), ("mapping", r"mutation mapping ([0-9.]+) s")):
        m = re.search(pat, line)
        if m:
            acc[key] = acc.get(key, 0.0) + float(m.group(1))
print(sys.argv[1:], "build_topology_s %.1f trees %d" % (d["build_topology_s"], d["trees"]), "sum over sections:", acc,
      "gpu ms/tree", d.get("gpu_builder_ms_per_tree"))
