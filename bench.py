#!/usr/bin/env python3
"""bench.py -- Paint (stepping-stone Li-Stephens forward/backward) throughput.

One "step" = one pass of the Paint hot path over one synthetic chunk already
resident in HBM: the backward and the forward pass for all N targets, one
launch of 2N workgroups (replacing the loop of pipeline/Paint.cpp:81-87 in the
reference).

    python bench.py [--gpus N --steps K --warmup W] [--haplotypes 5000 --snps 500000]

With --gpus N > 1 there is one rank per GPU: launched by torch.distributed.run (the
driver's way), or -- when no launcher set RANK -- by bench.py itself as a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`.  Ranks paint independent chunks (chunks are embarrassingly parallel in the
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


def cpu_baseline(N, L, bits, r, rpos, wb, seconds_target=12.0):
    """the oracle (plain-C port of the reference's PaintSteppingStones) timed on
    the host cores on a bounded sample of targets of the SAME chunk: on all cores
    (targets split over threads) and on ONE thread (the reference is
    single-threaded; SURVEY.md 8d asks for both)"""
    import rlutil
    o = rlutil.oracle()
    seq = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :N]
    seq = np.ascontiguousarray(seq + ord("0"), dtype=np.uint8)
    d = rlutil.RoData(N, L, seq.ctypes.data, r.ctypes.data, rpos.ctypes.data, 0.001)
    # ~3.1e8 updates/s/core measured on the reference; one target
    # costs 2*N*D_k ~ 2*N*0.11*L updates
    per_target = 2.0 * N * 0.12 * L / 3.0e8

    def sample(cores, seconds):
        per_thread = max(1, int(seconds / max(per_target, 1e-3)))
        count = min(N, cores * per_thread)
        stride = max(1, N // count)
        t0 = time.time()
        sites = o.ro_paint_sample(C.byref(d), wb.ctypes.data_as(C.c_void_p), len(wb) - 1, 0, stride, count, cores)
        dt = time.time() - t0
        assert sites > 0
        return dict(value=2.0 * N * sites / dt, unit="updates/s", cores=cores, kind="port",
                    sample="%d of %d targets (every %d-th) of the same chunk, oracle ro_paint_sample on %d "
                           "thread%s, %.1f s wall" % (count, N, stride, cores, "s" if cores > 1 else "", dt))

    port = sample(min(os.cpu_count() or 1, 64), seconds_target)
    port["single_thread"] = sample(1, seconds_target)
    # The REFERENCE itself where its build travelled with the snapshot (oracle/_ref, compiled from /root/reference by
    # oracle/Makefile in the build container): FastPainting::PaintSteppingStones of a few targets of the same chunk on
    # one thread (the reference's Paint is single-threaded), the harness's start-up (reading the 2.5 GB .hap) timed by
    # itself and taken off.
    out = port
    try:
        ref = reference_baseline(N, L, seq, r, rpos, wb, o, d, per_target, seconds_target)
        if ref:
            ref["port"] = port
            out = ref
    except Exception as e:
        port["reference_error"] = str(e)[:160]
    try:  # (a container may grant the process fewer cores' worth of CPU than it shows: the threads above share them)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            out["cpu_quota_cores"] = float(quota) / float(period)
    except Exception:
        pass
    return out


def reference_baseline(N, L, seq, r, rpos, wb, o, d, per_target, seconds_target):
    import subprocess
    import tempfile
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return None
    from relate_amd import api
    lib = api.lib()
    count = max(2, min(8, int(seconds_target / max(per_target, 1e-3))))
    stride = max(1, N // count)
    targets = [i * stride for i in range(count)]
    with tempfile.TemporaryDirectory() as work:
        dd = os.path.join(work, "out")
        os.makedirs(dd)
        W = len(wb) - 1
        bp = (np.arange(L, dtype=np.int64) * 100).astype(np.int32)  # (positions are not read by the painting)
        wbf = np.zeros(L + 2, dtype=np.int32)
        wbf[:W + 1] = wb
        lib.rl_write_chunk_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        assert lib.rl_write_chunk_files(dd.encode(), 0, N, L, seq.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p),
                                        r.ctypes.data_as(C.c_void_p), rpos.ctypes.data_as(C.c_void_p),
                                        wbf.ctypes.data_as(C.c_void_p), W) == 0

        def timed(ks):
            t0 = time.time()
            subprocess.run([harness, "paint_targets", "out", "0", "dump.bin"] + [str(k) for k in ks], cwd=work, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
            return time.time() - t0
        t_load = timed([])
        t_all = timed(targets)
    # the sites those targets visit (the oracle's count of the same plan)
    sites = o.ro_paint_sample(C.byref(d), wb.ctypes.data_as(C.c_void_p), len(wb) - 1, 0, stride, count, count)
    if sites <= 0:
        return None
    dt = max(t_all - t_load, 1e-6)
    return dict(value=2.0 * N * sites / dt, unit="updates/s", cores=1, kind="reference",
                sample="%d of %d targets (%s) of the same chunk through the unmodified reference's "
                       "FastPainting::PaintSteppingStones (oracle/_ref/ref_harness paint_targets, one thread): %.1f s, "
                       "of which %.1f s start-up (chunk files read) taken off" % (len(targets), N, targets, t_all, t_load))


def chunk_wallclock_full():
    """The metric's other half MEASURED: the whole C3 chunk -- Paint + BuildTopology of all 267 sections, files in ->
    .anc / .mut files out -- through `Relate --mode PaintBuildTopology` in a child process (tools/chunk_c3_fused.py).
    BENCH_C3_RUNS runs (default 2 -- a third would take the default bench past ten minutes; BENCH_C3_RUNS=3 for
    minimum / median / maximum; ~3 minutes each with the chunk's generation; profiles/r05_c3_runs.json holds sixty): the first with the stage's timing lines (RELATE_AMD_TIMING:
    trees built, RePaint busy seconds, `stage_lines`), the others without; every run's seconds are reported.  Before
    this process touches the GPU."""
    import statistics
    import subprocess
    tool = os.path.join(ROOT, "tools", "chunk_c3_fused.py")
    runs = max(1, int(os.environ.get("BENCH_C3_RUNS", "2")))
    results = []
    try:
        for i in range(runs):
            env = dict(os.environ)
            if i > 0:
                env["C3_NO_TIMING"] = "1"
            else:
                env.pop("C3_NO_TIMING", None)
            p = subprocess.run([sys.executable, tool, "267"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500,
                               env=env)
            results.append(json.loads(p.stdout.decode().strip().split("\n")[-1]))
        d = results[0]
        walls = [x["wall_s"] for x in results]
        checks = [verify_c3(x.get("section_md5", {})) for x in results]
        out = {"workload": "the C3 chunk, synthetic N=5000 x L=500000, %d windows: Relate --mode PaintBuildTopology, "
                           "Paint + BuildTopology of all %d sections in one process on one GPU, chunk files in, "
                           ".anc/.mut files out (no paint files)" % (d["windows"], d["sections"]),
               "measured": True, "wall_s": statistics.median(walls), "runs_s": walls, "min_s": min(walls),
               "median_s": statistics.median(walls), "max_s": max(walls), "runs": "run 0 with the stage's timing lines, the others without",
               "sections": d["sections"], "trees_kept": d["trees_kept"], "trees_built": d.get("trees_built"),
               "trees_built_per_s": (d["trees_built"] / d["wall_s"]) if d.get("trees_built") else None,
               "trees_kept_per_s": d["trees_kept"] / statistics.median(walls), "anc_GB": d["anc_GB"],
               "stage_lines": d.get("stage_lines", [])[:6]}
        out.update(checks[0])
        out["every_run_matches_reference"] = all(c.get("matches_reference") for c in checks)
        return out
    except Exception as e:  # never a reason to lose the bench line
        return {"error": str(e)[:200], "runs_s": [x.get("wall_s") for x in results]}


def verify_c3(section_md5):
    """This run's md5 of the C3 chunk's sections 0 / 133 / 266 against the REFERENCE's: tests/golden/c3_full.npz
    (section 133) and tests/golden/c3_ends.npz (the boundary sections 0 and 266) hold the md5 of out_<w>.anc /
    out_<w>.mut as the unmodified reference binary wrote them -- its own paint file of the window from
    PaintSteppingStones at full length, then Relate --mode BuildTopology of that section (tools/make_golden_c3.py,
    tools/make_golden_full.py)."""
    out = {"section_md5": section_md5, "verified_sections": [], "matches_reference": False}
    try:
        want = {}
        z = np.load(os.path.join(ROOT, "tests", "golden", "c3_full.npz"))
        w = int(z["pin_window"][0])
        want[w] = (z["w/anc_md5"].tobytes().hex(), z["w/mut_md5"].tobytes().hex())
        ze = np.load(os.path.join(ROOT, "tests", "golden", "c3_ends.npz"))
        for s in (int(x) for x in ze["sections"]):
            want[s] = (ze["s%d/anc_md5" % s].tobytes().hex(), ze["s%d/mut_md5" % s].tobytes().hex())
        out["reference_held_sections"] = sorted(want)
        for s, (anc, mut) in sorted(want.items()):
            if section_md5.get("out_%d.anc" % s) == anc and section_md5.get("out_%d.mut" % s) == mut:
                out["verified_sections"].append(s)
        out["matches_reference"] = out["verified_sections"] == sorted(want)
        out["verified_against"] = ("sections %s: the unmodified reference binary's .anc / .mut (tests/golden/c3_full.npz, "
                                   "c3_ends.npz)" % sorted(want))
    except Exception as e:
        out["verify_error"] = str(e)[:120]
    return out


def sample_against_reference(md5s):
    try:
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "n5000_l20k_ref.json")))
        return {"matches_reference": all(md5s.get(k) == v for k, v in ref["md5"].items()),
                "reference_seconds_one_thread": {"paint": ref["reference_paint_s"],
                                                 "build_topology_section_0": ref["reference_build_topology_section_0_s"]}}
    except Exception as e:
        return {"matches_reference": None, "reference_error": str(e)[:120]}


def chunk_wallclock_sample(sections=8, host_builder=False):
    """The other half of BASELINE.json's metric, "chunk wall-clock, N=5000": a bounded sample through the drop-in
    CLI, files in -> files out, in a child process (tools/chunk_wallclock_big.py): the Paint stage of an N=5000 x
    L=20000 chunk and BuildTopology of its first `sections` sections in ONE call, as the reference's scripts hand
    section ranges to a process (RelateParallel.sh:231-257) -- with several sections the trees are built on the GPU,
    a workgroup per tree, the sections side by side.  Not part of the timed steps."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "chunk_wallclock_big.py")
    try:
        # host_builder: the same sections with the trees built by the host's MinMatch (threaded, pinned to L3 groups)
        # instead of the device workers -- the CPU figure beside the chunk wall-clock
        env = dict(os.environ, RELATE_AMD_GPU_BUILD="0") if host_builder else dict(os.environ)
        p = subprocess.run([sys.executable, tool, "5000", "20000", "20", str(sections)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900, env=env)
        d = json.loads(p.stdout.decode().strip().split("\n")[-1])
        return {"workload": "synthetic N=5000 x L=20000 chunk (%d windows) through Relate --mode Paint (whole chunk) "
                            "and --mode BuildTopology (sections 0-%d in one call), chunk files in, paint/.anc/.mut "
                            "files out" % (d["windows"], d["sections_timed"] - 1),
                "paint_stage_s": d["paint_stage_s"], "paint_files_GB": d["paint_files_GB"],
                "build_topology_s": d["build_topology_s"], "sections": d["sections_timed"], "trees": d["trees"],
                "trees_per_s": d["trees"] / d["build_topology_s"],
                "snps_in_sections": d["snps_in_timed_sections"], "host_threads": d["host_threads"],
                "gpu_builder_ms_per_tree": d.get("gpu_builder_ms_per_tree"),
                "build_topology_phases": d.get("build_topology_phases", [])[:1],
                # window 0's paint file and section 0's .anc / .mut against the unmodified reference binary's on
                # the same chunk (tests/golden/n5000_l20k_ref.json: its md5s and its seconds in the build container)
                "md5": {k: v for k, v in d["md5"].items() if k.startswith("paint/") or "_0." in k},
                **sample_against_reference(d["md5"])}
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
    ap.add_argument("--mode", default="exact", choices=["exact", "lanes", "lanes32"])
    ap.add_argument("--skip-cpu", dest="no_cpu", action="store_true")
    ap.add_argument("--skip-alt", dest="no_alt", action="store_true", help="skip timing the other summation mode")
    ap.add_argument("--skip-k23", dest="skip_k23", action="store_true", help="skip the RePaint / matrix measurement")
    ap.add_argument("--skip-chunk", dest="skip_chunk", action="store_true",
                    help="skip the chunk wall-clock sample through the CLI (N=5000 runs on one GPU only)")
    ap.add_argument("--skip-full-chunk", dest="skip_full", action="store_true",
                    help="skip the whole-C3-chunk wall-clock (Paint + all 267 sections, ~3 minutes)")
    ap.add_argument("--skip-host-builder", dest="skip_host", action="store_true",
                    help="(the default since round 5: the whole-chunk runs take the time) skip the host-builder run "
                         "of the chunk sample")
    ap.add_argument("--host-builder", dest="with_host", action="store_true",
                    help="also run the 8-section chunk sample with the trees built by the host's threaded MinMatch (the "
                         "CPU figure beside the chunk wall-clock, ~2 minutes; round 5: 94.3 s against 58.6 s on the device, "
                         "profiles/r05_bench_c3_before_stripless_parts.json)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4"],
                    help="c3 (default; the configuration BASELINE.json's metric is quoted on): one chunk of N=5000 x "
                         "L=500k per GPU.  c4 (BASELINE.json config #4): N=2000 x 5M SNPs cut into ~50 chunks of "
                         "~121k SNPs (--memory 1), dealt to the GPUs round-robin as relate_amd.dist.run_chunks deals "
                         "them; each GPU paints --chunks-per-gpu of them per step")
    ap.add_argument("--chunks-per-gpu", dest="cpg", type=int, default=2)
    ap.add_argument("--print-launch", dest="print_launch", action="store_true",
                    help="with --gpus N > 1 and no launcher: print the launch command instead of running it")
    ap.add_argument("--shard", default="chunks", choices=["chunks", "targets"],
                    help="chunks (default, the contract's weak scaling): one chunk per GPU, no collective. "
                         "targets: ONE chunk for all ranks, each paints a range of target haplotypes (strong "
                         "scaling; BASELINE.json config #5) and the ranks all-gather one distance matrix's rows")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU) as a CHILD --
        # before this process has touched the GPU, and never by replacing it -- and pass rank 0's JSON line through.
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
        if args.print_launch:
            have = args.gpus
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d but %d GPU%s visible" % (args.gpus, have, "" if have == 1 else "s"))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
              [a for a in sys.argv[1:] if a != "--print-launch"]
        if args.print_launch:
            print(json.dumps(cmd))
            sys.exit(0)
        sys.exit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the chunk wall-clock sample runs in child processes BEFORE this process touches the GPU: the tree builder's
    # long-running workgroups are time-sliced against every other process that holds hardware queues on the device
    chunk_sample = chunk_full = chunk_host = None
    if world == 1 and args.n == 5000 and args.workload == "c3" and not args.skip_chunk:
        chunk_sample = chunk_wallclock_sample()
        if args.with_host and not args.skip_host:
            chunk_host = chunk_wallclock_sample(host_builder=True)
        if not args.skip_full:
            chunk_full = chunk_wallclock_full()
    import torch
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU, RCCL for the bookkeeping
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    from relate_amd import api
    from relate_amd import dist as rdist
    if args.workload == "c4":
        args.n, args.l, args.memory = 2000, 121000, 1.0
        args.skip_k23 = args.skip_chunk = True
    N, L = args.n, args.l
    by_target = args.shard == "targets"
    dev = local_rank if dist is not None else 0
    # the chunks of this rank: its own chunk (c3), or its deal of the job's chunk list (c4: chunk ids
    # 0 .. world * chunks_per_gpu - 1 dealt round-robin, relate_amd.dist.shard -- no data-path collective)
    my_chunks = [rank] if args.workload == "c3" else rdist.shard(list(range(world * args.cpg)), rank, world)
    ctxs = []
    for c in my_chunks:
        bits, r, rpos, wb = make_chunk(N, L, seed=1 if by_target else 1 + c, memory_gb=args.memory)
        cx = api.Context(dev)
        cx.set_chunk_bits(N, bits, r, rpos, wb)
        if by_target:
            cx.set_target_range(*rdist.target_range(rank, world, N))
        cx.prepare()  # plan on the host, panel + plan uploaded, stone buffers allocated: inputs resident in HBM
        ctxs.append(cx)
    ctx = ctxs[0]
    sites = sum(cx.total_sites() for cx in ctxs)
    updates = 2.0 * N * sites
    MODES = {"exact": api.RL_SUM_EXACT, "lanes": api.RL_SUM_LANES, "lanes32": api.RL_SUM_LANES32}
    mode = MODES[args.mode]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(m, steps, warmup):
        """-> wall seconds of `steps` Paints, mean duration of the launch in ms (HIP events on the launch's
        stream, taken inside rl_paint)"""
        for _ in range(warmup):
            for cx in ctxs:
                cx.paint(m)
        barrier()
        t0 = time.time()
        kern = 0.0
        for _ in range(steps):
            for cx in ctxs:
                kern += cx.paint(m)  # returns after the launch completed (HIP events)
        barrier()
        return time.time() - t0, kern / steps

    def split_times(m):
        """one launch per direction: (forward ms, backward ms) -- what each pass costs when it has the chip alone"""
        f = b = 0.0
        for cx in ctxs:
            cx.set_paint_split(True)
            cx.paint(m)
            ff, bb = cx.paint_times()
            cx.set_paint_split(False)
            f += ff
            b += bb
        return f, b

    dt, kernel_ms = timed(mode, args.steps, args.warmup)
    total_updates, dt = rdist.job_stats(updates, dt)  # sum of units, max of seconds over ranks
    fwd_ms, bwd_ms = split_times(mode) if rank == 0 else (0.0, 0.0)

    # the other summation modes on the same chunk (config.other_modes): `lanes` (re-associated sums, FP64 state) and
    # `lanes32` (the fast mode: packed-FP32 state, paint32_kernels.hip) next to the bit-exact default.  Useful
    # instructions per pair of directional updates: 9 FP64 (exact, lanes) / 5 with two donors per packed instruction.
    alts = {}
    fast_report = None
    try:  # what the -m gpu tests measured for the fast modes against the REFERENCE (tests/bigtile.py record)
        import glob
        fast_report = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fast_modes.json")))[-1]))
    except Exception:
        pass
    if not args.no_alt:
        for name, other in MODES.items():
            if other == mode:
                continue
            adt, ak = timed(other, max(1, args.steps), 1 if other != api.RL_SUM_EXACT else 0)
            af, ab = split_times(other) if rank == 0 else (0.0, 0.0)
            alts[name] = dict(mode=name, value=updates * max(1, args.steps) / adt,
                              ms_per_step=1e3 * adt / max(1, args.steps), kernel_ms=ak, fwd_alone_ms=af, bwd_alone_ms=ab,
                              roofline_frac=(2.0 * N * sites / 8.0) / (ak * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              bit_identical_to_reference=(name == "exact"))
            # the resource that binds is VALU issue (one instruction per lane and cycle, packed or not): useful
            # instructions per pair of directional updates -- 9 on doubles, 5 with two donors per packed instruction
            useful_n = 5.0 if name == "lanes32" else 9.0
            alts[name]["valu_issue"] = {"useful_instr_per_update_pair": useful_n,
                                        "achieved_Tinstr_per_s": useful_n * N * sites / (ak * 1e-3) / 1e12,
                                        "peak_Tinstr_per_s": FP64_PEAK_TINSTR,
                                        "frac": useful_n * N * sites / (ak * 1e-3) / 1e12 / FP64_PEAK_TINSTR}
            if fast_report and name in ("lanes", "lanes32"):
                alts[name]["against_reference_at_N5000"] = {
                    k[len("n5000_" + name + "_"):]: v for k, v in fast_report.items() if k.startswith("n5000_" + name + "_")}
    alt = alts.get("lanes") or alts.get("exact")

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
    roofline_k2 = None
    pmc = None
    pmc_file = None
    for cand in ("r06_pmc_c3.json", "r05_pmc_c3.json", "r04_pmc_c3.json", "r03_pmc_c3.json", "r02_pmc_c3.json"):  # HBM bytes per launch: separate rocprofv3 --pmc runs
        try:                                              # (tools/gpu_r06_i.sh), committed summary
            pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
            if pmc["N"] != N or pmc["L"] != L or by_target or args.workload != "c3":
                pmc = None
            else:
                pmc_file = "profiles/" + cand
                break
        except Exception:
            pmc = None
    if rank == 0 and not args.skip_k23 and not by_target:
        try:
            w = (len(wb) - 1) // 2
            ctx.open_window(w, None, int(wb[w]), mode).close()   # first launch loads the kernel's code object
            win = ctx.open_window(w, None, int(wb[w]), mode)     # stones resident after the last paint
            win.matrix(int(wb[w]))
            first_ms = win.matrix_ms
            # K3 through the ABI as the stage uses it (rows to a device buffer), mean of 20 calls: the first call on an
            # idle stream carries ~0.4 ms of launch / clock ramp that no later one sees
            dbuf = torch.empty((N, N), dtype=torch.float32, device="cuda")
            k3 = []
            for _ in range(20):
                win.matrix_rows_into(int(wb[w]), dbuf.data_ptr())
                k3.append(win.matrix_ms)
            win.matrix_ms = sum(k3) / len(k3)
            del dbuf
            rows = float(sum(win.rows(n) for n in range(N)))      # (target, visited site) pairs of the window
            W = len(wb) - 1
            # K2's algorithmic bytes per (target, visited site, donor): 4 B of posterior written + 2 bits of
            # panel read (forward and backward pass), SURVEY.md 8d
            k2_bytes = rows * N * 4.25
            k2_gbs = k2_bytes / (win.repaint_ms * 1e-3) / 1e9
            roofline_k2 = {"bound": "hbm", "kernel": "repaint_fwd_kernel + repaint_bwd_kernel (one window, all targets)",
                           "achieved": k2_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2_gbs / HBM_PEAK_GBS,
                           "traffic": pmc["kernels"]["repaint"]["hbm_bytes_per_launch"] if pmc else None,
                           "traffic_static_from": pmc_file if pmc else None,
                           "algorithmic_bytes": k2_bytes, "window": w, "target_site_rows": rows, "ms": win.repaint_ms}
            extra = {"window": w,
                     "k2_repaint_ms": win.repaint_ms,
                     "k2_topology_write_GBps": rows * N * 4.0 / (win.repaint_ms * 1e-3) / 1e9,
                     "k3_matrix_ms": win.matrix_ms,
                     "k3_matrix_first_call_ms": first_ms,
                     "k3_GBps": 12.0 * N * N / (win.matrix_ms * 1e-3) / 1e9,
                     # GPU time of one chunk of this shape: K1 once, K2 once per window (more with bounded
                     # windows), K3 once per tree built
                     "gpu_time_per_chunk": {"k1_paint_s": kernel_ms * 1e-3, "k2_all_windows_s": W * win.repaint_ms * 1e-3,
                                            "k3_per_tree_ms": win.matrix_ms, "windows": W}}
            win.close()
        except Exception as e:  # never let the secondary measurement break the bench line
            extra = {"error": str(e)[:200]}

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        # dominant kernel: paint_kernel, both directions of every target in one launch.  Algorithmic bytes =
        # 1 bit per directional update: 2 * N * sum_k D_k / 8 per launch.
        alg_bytes = 2.0 * N * sites / 8.0
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch: PMC passes (FETCH_SIZE + WRITE_SIZE) are separate rocprofv3 runs
        # (tools/gpu_profile_r05.sh); the committed summary is read here, NOT measured in this run
        traffic = pmc["kernels"][args.mode]["hbm_bytes_per_launch"] if pmc and args.mode in pmc["kernels"] else None
        traffic_from = pmc_file if traffic is not None else None
        # SURVEY.md 8d caveat H5: at 1 bit per update the FP64 vector pipe, not HBM, is the resource that binds.
        # USEFUL f64 instructions per pair of directional updates (one donor at one visited site, both passes):
        # 3 forward (add, masked mul, add into the sum) + 6 backward (masked add, add, masked mul, two for the
        # weighted term, add into the sum) = 9; whatever else the exact order costs is overhead, not counted.
        useful = 9.0 * N * sites / (kernel_ms * 1e-3) / 1e12
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
                             else ("%d chunk%s per GPU%s" % (len(my_chunks), "s" if len(my_chunks) > 1 else "",
                                                            " (config #4's shape: 5M SNPs = ~50 such chunks, dealt "
                                                            "round-robin)" if args.workload == "c4" else "")),
                             len(wb) - 1, args.memory),
                "sum_mode": args.mode,
                "sum_k_D_k": int(sites),
                "updates_per_step_per_gpu": updates,
                "nominal_updates_per_s_2NNL": 2.0 * N * N * L * world / (dt / args.steps),
                "kernel_ms": kernel_ms,
                "fwd_alone_ms": fwd_ms,
                "bwd_alone_ms": bwd_ms,
                "other_mode": alt,
                "other_modes": alts,
                "repaint_and_matrix": extra,
                "shard": args.shard,
                "target_shard_matrix": gather,
            },
            # (`bound`: the roofline the contract prices this path against -- HBM reads, BASELINE.json's north star -- and
            #  what achieved / peak / frac are quoted on; `binds`: what actually limits the kernel, FP64 VALU issue, with
            #  its own fraction under `fp64_valu`)
            "roofline": {"bound": "hbm", "binds": "fp64_valu", "kernel": "paint_kernel (forward + backward, one launch)",
                         "algorithmic_bytes": alg_bytes, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_static_from": traffic_from,
                         "fp64_valu": {"useful_instr_per_update_pair": 9,
                                       "achieved_Tinstr_per_s": useful,
                                       "peak_Tinstr_per_s": FP64_PEAK_TINSTR,
                                       "frac": useful / FP64_PEAK_TINSTR}},
        }
        if roofline_k2 is not None:
            out["roofline_k2"] = roofline_k2
        if not args.no_cpu and world == 1:  # (the CPU baseline is a one-GPU-run item: rank 0 at N=1 only)
            out["cpu_baseline"] = cpu_baseline(N, L, bits, r, rpos, wb)
        if chunk_sample is not None:
            out["config"]["chunk_wallclock_sample"] = chunk_sample
            if chunk_full is not None:
                out["config"]["chunk_wallclock_c3"] = chunk_full
            if chunk_host is not None and "cpu_baseline" in out:
                # the same 8 sections with the trees built on the host's cores (this library's threaded MinMatch;
                # the reference binary builds section 0 alone in 1211 s, DESIGN_NOTES.md 6)
                out["cpu_baseline"]["chunk_wallclock_sample_host_builder"] = {
                    k: chunk_host.get(k) for k in ("build_topology_s", "sections", "trees", "trees_per_s", "host_threads",
                                                   "error") if k in chunk_host}
        print(json.dumps(out), flush=True)
    for cx in ctxs:
        cx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
