# Round 5, run O: the woven matrix with its rows in blocks (MM_RBLOCK = 128 / 32: a merge touches ~70 / ~100 pages of
# 2 MB instead of ~390) -- variants of the library preloaded under the CLI.  The 8-section sample (a tree's phases with
# few trees in flight, the reference's md5s), then the whole C3 chunk, alternating with the default.
export TMPDIR=/tmp
O=gpurun_out/r05o
mkdir -p $O
V=$PWD/relate_amd/variants
for v in rb128 rb32; do
  LD_PRELOAD=$V/librelate_amd_$v.so RELATE_AMD_LIB=$V/librelate_amd_$v.so timeout 400 python tools/chunk_wallclock_big.py 5000 20000 20 8 > $O/s8_$v.json 2> $O/s8_$v.err; echo rc=$?
done
LD_PRELOAD=$V/librelate_amd_rb128.so timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_rb128_1.json 2> $O/e1.err; echo rc=$?
timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_default_1.json 2> $O/e2.err; echo rc=$?
LD_PRELOAD=$V/librelate_amd_rb128.so timeout 600 python tools/chunk_c3_fused.py 267 > $O/c3_rb128_2.json 2> $O/e3.err; echo rc=$?
python - <<'PY'
import json
ref=json.load(open("tests/golden/n5000_l20k_ref.json"))
for v in ("rb128","rb32"):
    try:
        d=json.load(open("gpurun_out/r05o/s8_%s.json"%v))
        print(v, round(d["build_topology_s"],1), "matches reference:", all(d["md5"].get(k)==x for k,x in ref["md5"].items()), d.get("gpu_builder_ms_per_tree"), d.get("builder_host_side",[])[:1])
    except Exception as e: print(v, "failed", e)
for f in ("c3_rb128_1","c3_default_1","c3_rb128_2"):
    try:
        d=json.load(open("gpurun_out/r05o/%s.json"%f))
        print(f, round(d["wall_s"],1), d.get("builder_host_side",[])[:1], d.get("stage_summary")[:1], d.get("section_md5",{}).get("out_133.anc"), d.get("per_window_mean_s"), d.get("per_section_mean_s"), d.get("gpu_builder_ms_per_tree"))
    except Exception as e: print(f, "failed", e)
PY
