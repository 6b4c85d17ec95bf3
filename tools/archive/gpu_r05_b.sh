# Round 5, run B: the whole C3 chunk at 124 and 132 workers (the per-tree kernels are lighter since the fusion: where is
# the cliff now?), the second with FindEquivalentBranches timed on the chunk's 22.9 GB of .anc files.
export TMPDIR=/tmp
O=gpurun_out/r05b
mkdir -p $O
RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124.json 2> $O/c3_w124.err; echo rc=$?
C3_FEB=1 RELATE_AMD_BUILD_WORKERS=132 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_w132.json 2> $O/c3_w132.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_w124","c3_w132"):
    try:
        d=json.load(open("gpurun_out/r05b/%s.json"%f))
        print(f, d["wall_s"], d.get("trees_built"), d.get("window_lines",[])[:2], d.get("builder_host_side",[])[:2], [l for l in d.get("stage_lines",[]) if l.startswith("[stage]")], d.get("find_equivalent_branches_s"), d.get("find_equivalent_branches_lines"))
    except Exception as e: print(f, "failed", e)
PY
