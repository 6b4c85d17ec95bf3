# Round 5, run I: adaptive workers with a remembered edge (96:116:140, 30 s hold) against fixed 116 on one box; the
# stress run of the mechanism; FindEquivalentBranches with per-thread workspaces, stand-alone and fused.
export TMPDIR=/tmp
O=gpurun_out/r05i
mkdir -p $O
RELATE_AMD_ADAPTIVE_WORKERS=1:8:16 RELATE_AMD_ADAPTIVE_HI=-1 RELATE_AMD_ADAPTIVE_LO=1000 timeout 600 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/s8_stress.json 2> $O/s8_stress.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05i/s8_stress.json")); ref=json.load(open("tests/golden/n5000_l20k_ref.json"))
print("stress:", d["build_topology_s"], "s, matches reference:", all(d["md5"].get(k)==v for k,v in ref["md5"].items()))
PY
grep -ac "sent home" $O/s8_stress.err
RELATE_AMD_ADAPTIVE_WORKERS=96:116:140 timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_adaptive_1.json 2> $O/c3_adaptive_1.err; echo rc=$?
C3_FEB=1 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_w116_feb.json 2> $O/c3_w116_feb.err; echo rc=$?
C3_FUSED_FEB=1 RELATE_AMD_ADAPTIVE_WORKERS=96:116:140 timeout 900 python tools/chunk_c3_fused.py 267 > $O/c3_adaptive_fused_feb.json 2> $O/c3_adaptive_fused_feb.err; echo rc=$?
python - <<'PY'
import json
for f in ("c3_adaptive_1","c3_w116_feb","c3_adaptive_fused_feb"):
    try:
        d=json.load(open("gpurun_out/r05i/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary"), d.get("sections_timeline",{}).get("sections_done_by_s"), d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"), d.get("find_equivalent_branches_s"), d.get("find_equivalent_branches_lines"), d.get("fused_feb_lines"), d.get("feb_md5"))
        print("   ", [l[23:] for l in d.get("builder_worker_launches",[]) if "waiting for RePaint" in l][:40])
    except Exception as e: print(f, "failed", e)
PY
