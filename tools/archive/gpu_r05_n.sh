# Round 5, run N: the strip-less part launches against the default at 116 workers, alternating, one box.
export TMPDIR=/tmp
O=gpurun_out/r05n
mkdir -p $O
for i in 1 2; do
  timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_$i.json 2> $O/a$i.err; echo rc=$?
  RELATE_AMD_REPAINT_NOSTRIP=1 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w116_nostrip_$i.json 2> $O/b$i.err; echo rc=$?
done
python - <<'PY'
import json
for f in ("c3_w116_1","c3_w116_nostrip_1","c3_w116_2","c3_w116_nostrip_2"):
    try:
        d=json.load(open("gpurun_out/r05n/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
    except Exception as e: print(f, "failed", e)
PY
