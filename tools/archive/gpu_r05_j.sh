# Round 5, run J: adaptive workers with and without RePaint's descent as a kernel of its own, alternating on one box.
export TMPDIR=/tmp
O=gpurun_out/r05j
mkdir -p $O
for i in 1 2; do
  RELATE_AMD_DESCENT_KERNEL=1 RELATE_AMD_ADAPTIVE_WORKERS=96:116:148 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_adaptive_descent_$i.json 2> $O/c3_adaptive_descent_$i.err; echo rc=$?
  RELATE_AMD_ADAPTIVE_WORKERS=96:116:148 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_adaptive_$i.json 2> $O/c3_adaptive_$i.err; echo rc=$?
done
python - <<'PY'
import json
for f in ("c3_adaptive_descent_1","c3_adaptive_1","c3_adaptive_descent_2","c3_adaptive_2"):
    try:
        d=json.load(open("gpurun_out/r05j/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("sections_timeline",{}).get("sections_done_by_s"), d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
        print("   ", [l[23:] for l in d.get("builder_worker_launches",[]) if "waiting for RePaint" in l][:30])
    except Exception as e: print(f, "failed", e)
PY
