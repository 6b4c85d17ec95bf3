#!/usr/bin/env python3
"""BASELINE.json config #5 at its FULL size as a job on one GPU: synthetic N = 10,000 x L = 200,000,
--memory 25, one chunk --
  (a) the fused stage, `Relate --mode PaintBuildTopology` of all sections (tools/chunk_c3_fused.py): the stepping stones
      (2 x 144 GB) are painted straight into pinned host memory, the windows are bounded, the trees are built by the
      device's workers (the 20-slot L_HOT kernel);
  (b) the same chunk through config #5's own route, relate_amd.dist.run_chunk_by_targets with ONE rank
      (tools/chunk_c5_sharded.py), for the first `route_sections` sections;
and from (a) the time the target-sharded route would take on 8 ranks, with its assumptions spelled out.

    python tools/c5_job_one_gpu.py [route_sections=48] [in_flight=48]   -> one JSON document on stdout"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, L, MEM = 10000, 200000, 25.0
route_sections = int(sys.argv[1]) if len(sys.argv) > 1 else 48
in_flight = int(sys.argv[2]) if len(sys.argv) > 2 else 48
out = {"N": N, "L": L, "memory": MEM}


def run(args, env=None, timeout=3000):
    t0 = time.time()
    p = subprocess.run([sys.executable] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    try:
        d = json.loads(p.stdout.decode().strip().split("\n")[-1])
    except Exception:
        d = {"error": p.stderr.decode()[-1500:], "rc": p.returncode}
    d["tool_wall_s"] = round(time.time() - t0, 1)
    return d


if os.environ.get("C5_SKIP_FUSED") != "1":
    out["fused_stage"] = run([os.path.join(ROOT, "tools", "chunk_c3_fused.py"), "99999", str(N), str(L), str(MEM)])
    print(json.dumps({"fused_stage": {k: out["fused_stage"].get(k) for k in ("wall_s", "sections", "trees_kept", "trees_built", "error")}}),
          file=sys.stderr, flush=True)
if route_sections > 0:
    out["by_targets_one_rank"] = run([os.path.join(ROOT, "tools", "chunk_c5_sharded.py"), str(N), str(L), str(MEM),
                                      str(route_sections), str(in_flight), os.environ.get("C5_ROUTE_FRACTION", "0.05"), "1"])
f = out.get("fused_stage") or {}
if f.get("wall_s"):
    trees = f.get("trees_built") or f.get("trees_kept")
    gather_bytes = 4.0 * N * N  # one all_gather_into_tensor of all ranks' row blocks per tree
    recv_per_rank = gather_bytes * 7.0 / 8.0
    xgmi = 7 * 50e9  # seven links per GPU; ~50 GB/s per link and direction sustained by RCCL's ring is the assumption
    out["projection_8_ranks"] = {
        "trees": trees,
        "all_gather_bytes_per_tree": gather_bytes,
        "received_per_rank_per_tree": recv_per_rank,
        "assumed_xgmi_receive_bandwidth_per_rank_Bps": xgmi,
        "all_gather_s_per_tree": recv_per_rank / xgmi,
        "all_gather_s_whole_job_per_rank": trees * recv_per_rank / xgmi,
        "compute_s_if_split_eightfold": f["wall_s"] / 8.0,
        "projected_job_s": max(f["wall_s"] / 8.0, trees * recv_per_rank / xgmi) + 5.0,
        "assumptions": "Paint, RePaint and the matrix rows split by target (1/8 each per rank); a section's trees are built "
                       "on its owner rank, the sections dealt evenly, so the trees split eightfold too; every tree costs one "
                       "all-gather of N^2 floats, overlapped with the other sections' builds -- the job takes the longer of "
                       "the two plus start-up.  Never run on more than one GPU from this environment."}
# section 0 against the REFERENCE where its fixture is there (tests/golden/c5_first.npz, tools/make_golden_full.py c5_first)
try:
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "c5_first.npz"))
    want = {"out_0.anc": z["s0/anc_md5"].tobytes().hex(), "out_0.mut": z["s0/mut_md5"].tobytes().hex()}
    for leg in ("fused_stage", "by_targets_one_rank"):
        got = (out.get(leg) or {}).get("section_md5") or (out.get(leg) or {}).get("md5") or {}
        if got:
            out[leg]["section_0_matches_reference"] = all(got.get(k) == v for k, v in want.items())
except Exception as e:
    out["reference_fixture"] = "not compared: %s" % str(e)[:100]
print(json.dumps(out))
