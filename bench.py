#!/usr/bin/env python3
"""bench.py -- Paint (stepping-stone Li-Stephens forward/backward) throughput.

One "step" = one pass of the Paint hot path over one synthetic chunk already
resident in HBM: the backward kernel + the forward kernel for all N targets
(replacing the loop of pipeline/Paint.cpp:81-87 in the reference).

    python bench.py [--gpus N --steps K --warmup W] [--haplotypes 5000 --snps 500000]

With --gpus N > 1 the driver launches one rank per GPU with torch.distributed.run;
ranks paint independent chunks (chunks are embarrassingly parallel in the
reference too: scripts/RelateParallel/RelateParallel.sh:216) -> weak scaling,
no data-path collective.  Rank 0 prints ONE JSON line.

Metric (BASELINE.json): directional haplotype-pair.SNP updates per second,
2 * N * sum_k D_k per Paint, D_k = sites visited by target k.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TINSTR = 39.3  # FP64 vector peak 78.6 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz x 2 for an FMA): 39.3e12 lane-instructions/s


def make_chunk(N, L, seed, memory_gb):
    """synthetic block-coalescent panel, bit-packed, + the reference's window rule"""
    from relate_amd import api
    lib = api.lib()
    rw = (N + 31) // 32
    bits = np.zeros((L, rw), dtype=np.uint32)
    r = np.zeros(L)
    rpos = np.zeros(L + 1)
    rc = lib.rl_synth_panel(N, L, C.c_uint64(seed), 100, 1, None, bits.ctypes.data_as(C.c_void_p), rw, None,
                            r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p))
    assert rc == 0
    # data.cpp:129: min_memory_size = mem*1e9/4 - (2N^2 + 3N) floats per window
    budget = memory_gb * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N)
    wbuf = np.zeros(L + 2, dtype=np.int32)
    W = lib.rl_synth_windows_bits(N, L, bits.ctypes.data_as(C.c_void_p), rw, C.c_double(budget),
                                  wbuf.ctypes.data_as(C.c_void_p), 499)
    assert W > 0, "window rule failed (raise --memory)"
    return bits, r, rpos, wbuf[:W + 1].copy()


def cpu_baseline(N, L, bits, r, rpos, wb, seconds_target=15.0):
    """the oracle (plain-C port of the reference's PaintSteppingStones) timed on
    the host cores on a bounded sample of targets of the SAME chunk"""
    import rlutil
    o = rlutil.oracle()
    cores = min(os.cpu_count() or 1, 64)
    seq = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :N]
    seq = np.ascontiguousarray(seq + ord("0"), dtype=np.uint8)
    d = rlutil.RoData(N, L, seq.ctypes.data, r.ctypes.data, rpos.ctypes.data, 0.001)
    # ~3.1e8 updates/s/core measured on the reference; one target
    # costs 2*N*D_k ~ 2*N*0.11*L updates
    per_target = 2.0 * N * 0.12 * L / 3.0e8
    per_thread = max(1, int(seconds_target / max(per_target, 1e-3)))
    count = min(N, cores * per_thread)
    stride = max(1, N // count)
    t0 = time.time()
    sites = o.ro_paint_sample(C.byref(d), wb.ctypes.data_as(C.c_void_p), len(wb) - 1, 0, stride, count, cores)
    dt = time.time() - t0
    assert sites > 0
    return dict(value=2.0 * N * sites / dt, unit="updates/s", cores=cores, kind="port",
                sample="%d of %d targets (every %d-th) of the same chunk, oracle ro_paint_sample on %d threads, "
                       "%.1f s wall" % (count, N, stride, cores, dt))


def chunk_wallclock_sample():
    """The other half of BASELINE.json's metric, "chunk wall-clock, N=5000": a bounded sample through the drop-in
    CLI, files in -> files out, in a child process (tools/chunk_wallclock_big.py): the Paint stage of an N=5000 x
    L=20000 chunk and BuildTopology of its first section.  Not part of the timed steps."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "chunk_wallclock_big.py")
    try:
        p = subprocess.run([sys.executable, tool, "5000", "20000", "20", "1"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900)
        d = json.loads(p.stdout.decode().strip().split("\n")[-1])
        return {"workload": "synthetic N=5000 x L=20000 chunk (%d windows) through Relate --mode Paint (whole chunk) "
                            "and --mode BuildTopology (section 0), chunk files in, paint/.anc/.mut files out" %
                            d["windows"],
                "paint_stage_s": d["paint_stage_s"], "paint_files_GB": d["paint_files_GB"],
                "build_topology_section_s": d["build_topology_s"], "trees": d["trees"],
                "snps_in_section": d["snps_in_timed_sections"], "host_threads": d["host_threads"],
                "build_topology_phases": d.get("build_topology_phases"), "md5": d["md5"]}
    except Exception as e:  # a sample, never a reason to lose the bench line
        return {"error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--haplotypes", dest="n", type=int, default=5000, help="haplotypes")
    ap.add_argument("--snps", dest="l", type=int, default=500000, help="SNPs")
    ap.add_argument("--memory", type=float, default=20.0, help="--memory of MakeChunks (window rule), GB")
    ap.add_argument("--mode", default="exact", choices=["exact", "lanes"])
    ap.add_argument("--skip-cpu", dest="no_cpu", action="store_true")
    ap.add_argument("--skip-alt", dest="no_alt", action="store_true", help="skip timing the other summation mode")
    ap.add_argument("--skip-k23", dest="skip_k23", action="store_true", help="skip the RePaint / matrix measurement")
    ap.add_argument("--skip-chunk", dest="skip_chunk", action="store_true",
                    help="skip the chunk wall-clock sample through the CLI (N=5000 runs on one GPU only)")
    ap.add_argument("--shard", default="chunks", choices=["chunks", "targets"],
                    help="chunks (default, the contract's weak scaling): one chunk per GPU, no collective. "
                         "targets: ONE chunk for all ranks, each paints a range of target haplotypes (strong "
                         "scaling; BASELINE.json config #5) and the ranks all-gather one distance matrix's rows")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU, RCCL for the bookkeeping
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    from relate_amd import api
    N, L = args.n, args.l
    by_target = args.shard == "targets"
    bits, r, rpos, wb = make_chunk(N, L, seed=1 if by_target else 1 + rank, memory_gb=args.memory)
    ctx = api.Context(local_rank if dist is not None else 0)
    ctx.set_chunk_bits(N, bits, r, rpos, wb)
    if by_target:
        from relate_amd import dist as rdist0
        ctx.set_target_range(*rdist0.target_range(rank, world, N))
    ctx.prepare()  # plan on the host, panel + plan uploaded, stone buffers allocated: inputs resident in HBM
    sites = ctx.total_sites()
    updates = 2.0 * N * sites
    mode = api.RL_SUM_EXACT if args.mode == "exact" else api.RL_SUM_LANES

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(m, steps, warmup):
        for _ in range(warmup):
            ctx.paint(m)
        barrier()
        t0 = time.time()
        fwd = bwd = 0.0
        for _ in range(steps):
            ctx.paint(m)  # returns after both kernels completed (HIP events)
            f, b = ctx.paint_times()
            fwd += f
            bwd += b
        barrier()
        return time.time() - t0, fwd / steps, bwd / steps

    dt, fwd_ms, bwd_ms = timed(mode, args.steps, args.warmup)
    from relate_amd import dist as rdist
    total_updates, dt = rdist.job_stats(updates, dt)  # sum of units, max of seconds over ranks

    alt = None
    if not args.no_alt:
        other = api.RL_SUM_LANES if mode == api.RL_SUM_EXACT else api.RL_SUM_EXACT
        adt, af, ab = timed(other, max(1, args.steps), 1 if other == api.RL_SUM_LANES else 0)
        alt = dict(mode="lanes" if other == api.RL_SUM_LANES else "exact",
                   value=updates * max(1, args.steps) / adt, ms_per_step=1e3 * adt / max(1, args.steps),
                   fwd_kernel_ms=af, bwd_kernel_ms=ab,
                   bwd_roofline_frac=(N * sites / 8.0) / (ab * 1e-3) / 1e9 / HBM_PEAK_GBS)

    # secondary kernels of the path on the same chunk (reported under config, not timed steps):
    # K2 RePaintSection of one window for all targets, K3 one N x N distance matrix
    gather = None
    if by_target:  # every rank: rows of one distance matrix -> all-gather (RCCL) -> N x N on every rank
        w = (len(wb) - 1) // 2
        win = ctx.open_window(w, None, int(wb[w]), mode)
        k0, k1 = ctx.target_range()
        rows = torch.empty((k1 - k0, N), dtype=torch.float32, device="cuda")
        win.matrix_rows_into(int(wb[w]), rows.data_ptr())
        barrier()
        t0 = time.time()
        full = rdist.all_gather_rows(rows, N)
        barrier()
        gather = {"window": w, "k2_repaint_ms": win.repaint_ms, "k3_matrix_ms": win.matrix_ms,
                  "all_gather_ms": 1e3 * (time.time() - t0), "matrix_shape": list(full.shape)}
        win.close()
    extra = None
    if rank == 0 and not args.skip_k23 and not by_target:
        try:
            w = (len(wb) - 1) // 2
            ctx.open_window(w, None, int(wb[w]), mode).close()   # first launch loads the kernel's code object
            win = ctx.open_window(w, None, int(wb[w]), mode)     # stones resident after the last paint
            win.matrix(int(wb[w]))
            rows = sum(win.rows(n) for n in range(0, N, 50)) * 50.0   # sampled row count
            extra = {"window": w,
                     "k2_repaint_ms": win.repaint_ms,
                     "k2_topology_write_GBps": rows * N * 4.0 / (win.repaint_ms * 1e-3) / 1e9,
                     "k3_matrix_ms": win.matrix_ms,
                     "k3_GBps": 12.0 * N * N / (win.matrix_ms * 1e-3) / 1e9}
            win.close()
        except Exception as e:  # never let the secondary measurement break the bench line
            extra = {"error": str(e)[:200]}

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        # dominant kernel: the backward launch.  Algorithmic bytes = 1 bit per
        # directional update: N * sum_k D_k / 8 per launch.
        alg_bytes = N * sites / 8.0
        achieved = alg_bytes / (bwd_ms * 1e-3) / 1e9
        # HBM bytes of that launch from the PMC passes (FETCH_SIZE + WRITE_SIZE, collected by
        # tools/gpu_profile_r01.sh in separate rocprofv3 runs, summary committed under profiles/)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_c3.json")))
            if pmc["N"] == N and pmc["L"] == L and not by_target:
                traffic = pmc["kernels"][args.mode + "_bwd"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "haplotype-pair*SNP updates/sec (Paint)",
            "value": total_updates * args.steps / dt,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if by_target else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic block-coalescent panel, N=%d haplotypes x L=%d SNPs, %s, "
                            "%d windows (--memory %g), Paint = PaintSteppingStones for all targets" %
                            (N, L, "ONE chunk sharded by target haplotype over the GPUs" if by_target
                             else "1 chunk per GPU", len(wb) - 1, args.memory),
                "sum_mode": args.mode,
                "sum_k_D_k": int(sites),
                "updates_per_step_per_gpu": updates,
                "nominal_updates_per_s_2NNL": 2.0 * N * N * L * world / (dt / args.steps),
                "fwd_kernel_ms": fwd_ms,
                "bwd_kernel_ms": bwd_ms,
                "other_mode": alt,
                "repaint_and_matrix": extra,
                "shard": args.shard,
                "target_shard_matrix": gather,
            },
            "roofline": {"bound": "hbm", "kernel": "paint_kernel<backward>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         # SURVEY.md 8d caveat H5: at 1 bit per update the FP64 vector pipe, not HBM, is the
                         # resource that binds.  Algorithmic f64 instructions per directional update of the
                         # backward kernel: 6 (update + weighted lane sum), + 7 for the exact sum.
                         "fp64_valu": {"instr_per_update": 13 if args.mode == "exact" else 6,
                                       "achieved_Tinstr_per_s": (13 if args.mode == "exact" else 6) * N * sites
                                       / (bwd_ms * 1e-3) / 1e12,
                                       "peak_Tinstr_per_s": FP64_PEAK_TINSTR,
                                       "frac": (13 if args.mode == "exact" else 6) * N * sites / (bwd_ms * 1e-3) / 1e12
                                       / FP64_PEAK_TINSTR}},
        }
        if not args.no_cpu and world == 1:  # (the CPU baseline is a one-GPU-run item: rank 0 at N=1 only)
            out["cpu_baseline"] = cpu_baseline(N, L, bits, r, rpos, wb)
        if world == 1 and N == 5000 and not args.skip_chunk:
            ctx.close()
            ctx = None
            out["config"]["chunk_wallclock_sample"] = chunk_wallclock_sample()
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
