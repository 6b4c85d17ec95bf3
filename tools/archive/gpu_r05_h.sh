# Round 5, run F: the worker count following RePaint's queue (RELATE_AMD_ADAPTIVE_WORKERS=lo:start:hi).  First a
# stress run of the mechanism on the 8-section sample (a launch sent home or eight workers added every 2 s whatever
# the queue says: same files?), then the whole C3 chunk: adaptive 96:116:148, fixed 116, adaptive again.
export TMPDIR=/tmp
O=gpurun_out/r05h
mkdir -p $O
RELATE_AMD_ADAPTIVE_WORKERS=1:8:16 RELATE_AMD_ADAPTIVE_HI=-1 RELATE_AMD_ADAPTIVE_LO=1000 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/s8_stress.json 2> $O/s8_stress.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05h/s8_stress.json")); ref=json.load(open("tests/golden/n5000_l20k_ref.json"))
print("stress:", d["build_topology_s"], "s, matches reference:", all(d["md5"].get(k)==v for k,v in ref["md5"].items()))
PY
grep -a "sent home\|sections waiting" $O/s8_stress.err | head -5; grep -ac "sent home" $O/s8_stress.err
RELATE_AMD_ADAPTIVE_WORKERS=96:116:148 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_adaptive_1.json 2> $O/c3_adaptive_1.err; echo rc=$?
RELATE_AMD_BUILD_WORKERS=124 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_w124.json 2> $O/c3_w124.err; echo rc=$?
RELATE_AMD_ADAPTIVE_WORKERS=96:116:148 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_adaptive_2.json 2> $O/c3_adaptive_2.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_adaptive_1","c3_w124","c3_adaptive_2"):
    try:
        d=json.load(open("gpurun_out/r05h/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("window_lines",[])[:2], d.get("builder_host_side",[])[:1], d.get("stage_summary"), d.get("sections_timeline",{}).get("sections_done_by_s"), d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"))
        print("   ", [l for l in d.get("builder_worker_launches",[]) if "waiting for RePaint" in l][:30])
    except Exception as e: print(f, "failed", e)
PY
