#!/usr/bin/env python3
"""The stage's worker rule (treeseq.cpp: stage_worker_goal) against other counts: the fused stage of
one chunk per tree size -- N = 2000 (a C4 chunk), N = 5000 (the L = 100k cut of C3, 53 windows), N = 10,000
(L = 20k, 33 windows) -- with the rule's own count and with RELATE_AMD_BUILD_WORKERS = 0.75 / 1.25 / 1.5 x of it.

    python tools/worker_rule_sweep.py [2000] [5000] [10000]  -> one JSON document on stdout"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {2000: (2000, 121000, 1.0), 5000: (5000, 100000, 20.0), 10000: (10000, 20000, 25.0)}
want = [int(x) for x in sys.argv[1:]] or [2000, 5000, 10000]
out = {}
for n in want:
    N, L, mem = CASES[n]
    runs = []

    def one(workers):
        env = dict(os.environ, RELATE_AMD_TIMING="1")
        if workers:
            env["RELATE_AMD_BUILD_WORKERS"] = str(workers)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "chunk_c3_fused.py"), "9999", str(N), str(L), str(mem)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
        d = json.loads(p.stdout.decode().strip().split("\n")[-1])
        goal = None
        for l in d.get("stage_lines", []) + d.get("worker_lines", []):
            m = re.search(r"goal (\d+)", l)
            if m:
                goal = int(m.group(1))
        return {"workers_env": workers, "wall_s": round(d["wall_s"], 2), "trees_built": d.get("trees_built"),
                "trees_kept": d.get("trees_kept"), "sections": d.get("sections"), "goal_seen": goal,
                "stage_summary": ([l for l in d.get("stage_summary", []) if "workers asked for" in l] or [""])[0][:400]}
    base = one(0)
    runs.append(base)
    m = re.search(r"on (\d+) threads, up to (\d+) open", base["stage_summary"] or "")
    rule = None
    for l in [base["stage_summary"]]:
        mm = re.search(r"\((\d+) workers asked for\)", l or "")
        if mm:
            rule = int(mm.group(1))
    out[str(n)] = {"N": N, "L": L, "memory": mem, "rule": base, "others": []}
    rule = rule or base.get("goal_seen")
    if rule:
        for f in (0.75, 1.25, 1.5):
            w = max(8, int(rule * f) // 8 * 8)
            out[str(n)]["others"].append(dict(one(w), factor=f))
        best = min([base["wall_s"]] + [x["wall_s"] for x in out[str(n)]["others"]])
        out[str(n)]["rule_workers"] = rule
        out[str(n)]["rule_over_best"] = round(base["wall_s"] / best, 3)
    print(json.dumps({str(n): out[str(n)]}), file=sys.stderr, flush=True)
print(json.dumps(out))
