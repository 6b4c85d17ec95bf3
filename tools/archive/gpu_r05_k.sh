# Round 5, run K: how much host CPU the stage burns (the box gives the process a quota) and whether blocking waits
# instead of spinning help; FindEquivalentBranches with the chain search, stand-alone and fused.
export TMPDIR=/tmp
O=gpurun_out/r05k
mkdir -p $O
cat /sys/fs/cgroup/cpu.max > $O/box.txt 2>&1; nproc >> $O/box.txt; cat /sys/fs/cgroup/cpu.stat >> $O/box.txt 2>&1
C3_FEB=1 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_default_feb.json 2> $O/c3_default_feb.err; echo rc=$?
RELATE_AMD_BLOCKING_SYNC=1 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_blocking.json 2> $O/c3_blocking.err; echo rc=$?
C3_FUSED_FEB=1 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_fused_feb.json 2> $O/c3_fused_feb.err; echo rc=$?
RELATE_AMD_BLOCKING_SYNC=1 RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_blocking_w124.json 2> $O/c3_blocking_w124.err; echo rc=$?
cat /sys/fs/cgroup/cpu.stat >> $O/box.txt 2>&1
cat $O/box.txt | head -20
python - <<'PY'
import json
for f in ("c3_default_feb","c3_blocking","c3_fused_feb","c3_blocking_w124"):
    try:
        d=json.load(open("gpurun_out/r05k/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("stage_cpu_s"), d.get("cgroup_cpu_max"), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"), d.get("find_equivalent_branches_s"), [l for l in (d.get("find_equivalent_branches_lines") or []) if "equivalent" in l or "CPU" in l], d.get("fused_feb_lines"), d.get("feb_md5"))
    except Exception as e: print(f, "failed", e)
PY
