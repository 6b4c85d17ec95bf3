#!/usr/bin/env python3
"""profiles/r05_c3_runs.json: every whole-C3-chunk run of round 5 (tools/chunk_c3_fused.py 267 on the GPU box, scripts
tools/gpu_r05_*.sh), the fields a reader needs: wall-clock, trees, worker count, what a window waited for RePaint, the
tree's phases under load, FindEquivalentBranches (stand-alone and fused) where it ran."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOTES = {"r05a": "per-tree kernels fused, step 1 (penalty + row minima, prior + row minima, weave + pair scan)",
         "r05b": "as r05a; FindEquivalentBranches stand-alone with one fread per field (the reference's pattern)",
         "r05c": "as r05a; FindEquivalentBranches with one read / write per tree, files on threads; fused behind the stage (pool of 32)",
         "r05d": "weave both ways (each element of D / CF read once) + K3 with penalty and row minima",
         "r05e": "as r05d", "r05f": "as r05d", "r05g": "as r05d"}
out = {"note": __doc__.strip(), "runs": {}}
for fn in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r05*", "c3_*.json"))):
    tag = os.path.relpath(fn, os.path.join(ROOT, "gpurun_out"))
    try:
        d = json.load(open(fn))
    except Exception:
        continue
    if "wall_s" not in d:
        continue
    m = re.search(r"_w(\d+)", tag)
    waits = [float(x) for l in d.get("window_lines", []) for x in re.findall(r"turn ([\d.]+)", l)]
    launches = [float(x) for l in d.get("window_lines", []) for x in re.findall(r"RePaint launches ([\d.]+)", l)]
    sub = [float(x) for l in d.get("builder_host_side", []) for x in re.findall(r"submit -> done ([\d.]+)", l)]
    stage = [l for l in d.get("stage_lines", []) if l.startswith("[stage] sections")]
    out["runs"][tag] = {
        "library": NOTES.get(tag.split("/")[0], ""), "workers": int(m.group(1)) if m else "default",
        "descent_kernel": "descent" in tag, "wall_s": round(d["wall_s"], 1), "trees_built": d.get("trees_built"),
        "trees_kept": d.get("trees_kept"), "repaint_wait_per_window_s": [round(min(waits), 1), round(max(waits), 1)] if waits else None,
        "repaint_launch_s_per_window": [round(min(launches), 2), round(max(launches), 2)] if launches else None,
        "tree_submit_to_done_ms": round(sum(sub) / len(sub), 1) if sub else None,
        "tree_phases_ms": d.get("gpu_builder_ms_per_tree"), "stage_line": stage[:1],
        "section_md5": d.get("section_md5"),
        "find_equivalent_branches_s": d.get("find_equivalent_branches_s"),
        "find_equivalent_branches_lines": [l for l in d.get("find_equivalent_branches_lines", []) if "find equivalent" in l or "CPU Time" in l],
        "fused_feb_lines": d.get("fused_feb_lines")}
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_c3_runs.json"), "w"), indent=1)
for k, v in out["runs"].items():
    print(k, v["wall_s"], v["workers"], v["repaint_wait_per_window_s"], v["tree_submit_to_done_ms"], v["find_equivalent_branches_s"])
