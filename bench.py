#!/usr/bin/env python3
"""bench.py — whole-job throughput of the SSDR-AL hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the full hot path over one batch of 16 synthetic S3DIS-like rooms per GPU, inputs already
resident in HBM:  grid-subsample (dl 0.04) of each raw room -> 40960-point tile around a picked centre -> 5-level
KNN pyramid (k 16) -> RandLA-Net inference (split-bf16 matrix products, fp32 everywhere else) -> point / superpoint
uncertainty, class balance, ranking ->
candidate features -> per-cloud chamfer graph + one propagation hop -> FPS selection.
Metric = input tile points through that whole pipe per second (BASELINE.json).  Tiles shard across GPUs with no
data-path collective until the candidates' propagated features are all-gathered before the (replicated) global
FPS: weak scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ssdr-al_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
PEAK_F64_VECTOR_TFLOPS = 78.6     # AMD's public MI355X figure for vector float64 (the guide lists none; measured issue rate: one v_fma_f64 per 5.4 cycles and SIMD = 58 TF)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
PROFILE_ROUND = "r06"             # profiles/<round>_pmc_summary.json supplies roofline.traffic (with its commit)
DTYPE = {"f32": "f32 (exact f32-input MFMA)",
         "bf16x3": "bf16x3: split-bf16 MFMA operands (hi*hi + lo*hi + hi*lo), fp32 accumulate, fp32 activations and non-matrix arithmetic",
         "bf16": "bf16 MFMA operands, fp32 accumulate, fp32 activations and non-matrix arithmetic"}
TILES_PER_GPU = 16
RAW_DENSITY = 5000.0              # points / m^2 -> 0.4-1.2 M raw points per room


def count_gpus():
    """GPU agents the kernel driver lists (KFD topology nodes with SIMDs), without touching HIP: the parent of an N-rank launch must not
    initialise a GPU, and a launch onto fewer devices than ranks must fail before any rank does."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        nodes = os.listdir(base)
    except OSError:
        return 0                      # no amdgpu compute driver on this node: no GPU
    for d in nodes:
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) >= 2)
        except OSError:
            return None               # the topology is there but not readable for this user: no verdict (the launch goes ahead)
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n):
    """One process per GPU on this node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same flags>` as a
    child process (never an exec: this process may not have touched the GPU, and it must stay that way until the child is started)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: > 1 s of timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run the batches strictly one after the other")
    ap.add_argument("--pipeline-depth", type=int, default=0, choices=(0, 2, 3, 4, 5),
                    help="batches in flight (stages on separate HIP streams); 0 = 5 on one GPU, 4 with the exchanges of N > 1")
    ap.add_argument("--global-batch", type=int, default=0, help="sampling()'s batch_size as ONE number for the job (the reference's 10 000 / 3 000 regions per round, "
                    "ssdr_main_S3DIS2.py:134) instead of 37 regions per tile: with N ranks the replicated global chain then stays the size of one rank's")
    ap.add_argument("--schedule", choices=("stage", "batch"), default=os.environ.get("SSDR_BENCH_SCHEDULE", "stage"),
                    help="batches in flight as a stream per STAGE (pipeline.Pipelined) or a stream per BATCH (pipeline.BatchStreams)")
    ap.add_argument("--slots", type=int, default=4, help="--schedule batch: batches (= streams) in flight")
    ap.add_argument("--sel-streams", type=int, default=0, help="--schedule batch: streams the selections take in turn (0: the batch's own stream)")
    ap.add_argument("--select-lag", type=int, default=1, help="selections in flight behind the newest one the host waits for (pipeline.Pipelined sel_lag; 1 = the previous batch's)")
    ap.add_argument("--spare-set", action="store_true", help="one more buffer set than batches in flight: no stage is deferred behind the wait for the previous selection (A/B timing)")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage timing line to stderr")
    ap.add_argument("--precision", default="bf16x3", choices=("f32", "bf16x3", "bf16"),
                    help="arithmetic of the network's matrix products (activations / accumulation are fp32 in every mode); bf16x3 = split bf16, "
                         "inside the 1e-3 feature tolerance of the fp32 path (tests/test_randla.py); bf16 = BASELINE configuration 3")
    ap.add_argument("--tiles16", action="store_true", help="bf16 modes: the 16 x 16-tile attention kernels of rounds 1-3 instead of the 32 x 32 formulation (A/B timing)")
    ap.add_argument("--selector", default="fps", choices=("fps", "kcenter"),
                    help="final selection over the (gathered) propagated features: FPS (the paper's gcn_fps branch) or the global k-center of BASELINE configuration 4")
    ap.add_argument("--no-al-round", action="store_true", help="skip the untimed-by-the-headline leg that runs ONE active-learning round at the reference's scale")
    ap.add_argument("--al-batches", type=int, default=17, help="batches of 16 tiles in that round (17 = the 272 rooms of S3DIS)")
    ap.add_argument("--al-batch-size", type=int, default=10000, help="sampling()'s batch_size in that round (ssdr_main_S3DIS2.py:134)")
    ap.add_argument("--emu", action="store_true",
                    help="TEST ONLY (tests/test_bench_launch.py): CPU logic build of the kernels + gloo, a tiny workload; exercises the launcher and the "
                         "N > 1 control flow on a box without GPUs, measures nothing")
    args = ap.parse_args()

    if args.gpus > 1 and not args.emu and not os.environ.get("SSDR_BENCH_SKIP_DEVICE_COUNT"):
        have = count_gpus()
        if have is not None and have < args.gpus:       # before any rank touches a GPU (every rank of a launcher's job checks for itself: same answer, same exit)
            raise SystemExit("bench.py: --gpus %d but this node lists %d GPU(s) (KFD topology / *_VISIBLE_DEVICES)" % (args.gpus, have))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) as CHILDREN, through
        # torch.distributed.run, before this process has made any GPU call; rank 0's JSON line comes through on the
        # inherited stdout and the exit code is non-zero if any rank failed.
        sys.exit(launch_ranks(args.gpus))

    # ONE JSON line on stdout and nothing else: native libraries in the process write there too (RCCL prints a five-line version banner through C stdio, which
    # a pipe holds back until exit — BEHIND the JSON line).  Everything that is not the line goes to stderr: descriptor 1 is pointed at stderr for the life of the
    # process and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (this pool's driver shares device memory between processes through dmabuf only: RCCL needs it; set before HIP starts)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("SSDR_BENCH_FORCE_DIST"):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher)" % (args.gpus, world, args.gpus))
    if os.environ.get("SSDR_AL_BENCH_FAIL_RANK") == str(rank):       # tests/test_bench_launch.py: a failing rank must fail the whole launch
        raise SystemExit("bench.py: rank %d told to fail" % rank)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("SSDR_BENCH_FORCE_DIST"))      # the latter: exercise the RCCL path on one GPU
    if args.pipeline_depth == 0:
        # One GPU: every stage on its own stream (99 vs 86 Mpoints/s for the 4-stream grouping on the same box).  With the
        # exchanges torch / RCCL bring their own streams into the process and the 4 hardware queues get shared: the grouping
        # front end + KNN | network | scoring | selection then wins by far (one GPU through RCCL: 92 vs 63 Mpoints/s).
        args.pipeline_depth = 4 if use_dist else 5
    if use_dist and args.emu:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    elif use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # the exchanges are small and host-synchronous: a high-priority RCCL stream gets a hardware queue of its own instead of
        # waiting behind the queued kernels of whichever pipeline stream it would otherwise share one with
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=not os.environ.get("SSDR_NCCL_NORMAL_PRIO"))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)

    ranks_seen = 1
    if use_dist:        # what the communicator itself saw: an all-reduce of ones (a SCALE record with n_gpus = N must also show N here)
        import torch
        one = torch.ones(1, dtype=torch.int32, device="cpu" if args.emu else torch.device("cuda", local_rank))
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    from ssdr_al import _lib, pipeline, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS
    Cfg, tiles_per_gpu, density, hp_kw = ConfigS3DIS, TILES_PER_GPU, RAW_DENSITY, {}
    if args.emu:
        _lib.use(os.path.join(ROOT, "tests", "hipemu", "libssdr_al_emu.so"))

        class Cfg(ConfigS3DIS):
            num_points = 1024
        tiles_per_gpu, density, hp_kw = 2, 80.0, dict(select_per_tile=5, labeled_per_tile=2)
        args.no_cpu_baseline = True
    if args.global_batch > 0:
        hp_kw = dict(hp_kw, batch_size=args.global_batch)
    _lib.check(_lib.lib().ssdr_init(0 if args.emu else local_rank))

    weights = synthetic.init_weights(0)         # random-init weights of the reference architecture (helper_tf_util.py:43-48 rule)
    ids = [rank * tiles_per_gpu + i for i in range(tiles_per_gpu)]
    rooms = [synthetic.make_room(5000 + i, density=density) for i in ids]

    def mk():
        return pipeline.HotPath(weights, Cfg, precision=args.precision, selector=args.selector, tiles32=not args.tiles16, **hp_kw).load_rooms(rooms, ids)
    hp = mk()

    gather = None
    if use_dist:
        from ssdr_al.distributed import Comm
        gather = Comm(dist, "cpu" if args.emu else "cuda")             # the three small exchanges of the selection stage (RCCL)

    # Consecutive batches are software-pipelined over the stages (front end | KNN pyramid | network | scoring | selection
    # on separate HIP streams, one buffer set per batch in flight).  N > 1: the three exchanges run on device buffers,
    # ordered on the same streams, in the same order on every rank.
    pipe = None
    if not args.no_pipeline:
        if args.schedule == "batch":
            pipe = pipeline.BatchStreams(mk, args.slots, args.sel_streams)
            args.pipeline_depth = args.slots
        else:
            pipe = pipeline.Pipelined(mk, args.pipeline_depth, sel_lag=args.select_lag, spare_set=args.spare_set)

    def barrier():
        # EVERYTHING issued so far has finished on this rank (all stage and selection streams, not only the library stream), then all ranks meet
        if pipe is not None:
            pipe.finish()
        _lib.sync()
        if use_dist:
            if not args.emu:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    # Timed region: K steps between two points at which the GPU is idle.  With batches in flight the region therefore holds the fill and
    # the drain of the pipe as well (the first steps find it empty, the last batches are completed before the clock stops): K launch
    # sequences of every stage, K completed selections, nothing left in flight at either end.
    if pipe is not None:
        pipe.run(max(args.warmup, 1), gather)
    else:
        for _ in range(args.warmup):
            hp.step(gather)
    barrier()
    t0 = time.perf_counter()
    if pipe is not None:
        pipe.run(args.steps, gather)
    else:
        for _ in range(args.steps):
            hp.step(gather)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        import torch
        t = torch.tensor([dt], device="cpu" if args.emu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    npts = world * tiles_per_gpu * Cfg.num_points * args.steps
    value = npts / dt / 1e6

    # N > 1: the second reading.  `value` keeps the bench's default, 37 regions per tile of the WHOLE job (the picks grow with N, so the replicated
    # global farthest-point chain is N x the picks over N x the rows: N^2); the reference's own reading is ONE batch_size per round whatever the
    # number of clouds (ssdr_main_S3DIS2.py:134) — the chain then stays what one rank runs.  Same path, same timed region, the other batch size.
    fixed_batch = None
    if use_dist and args.global_batch == 0 and (world > 1 or os.environ.get("SSDR_BENCH_FORCE_DIST")):
        K_fixed = hp.select_per_tile * tiles_per_gpu
        hps = [hp] + (list(pipe.hp) if pipe is not None else [])

        def set_batch(k):
            for h in hps:
                h.batch_size = k; h._dist = None; h._select_static()
        set_batch(K_fixed)
        if not args.emu:                              # (warm-up with the new tables; the CPU logic build measures nothing)
            if pipe is not None:
                pipe.run(max(args.warmup, 1), gather)
            else:
                hp.step(gather)
        barrier()
        steps2 = 1 if args.emu else args.steps          # (the CPU logic build measures nothing: one step exercises the control flow)
        t1 = time.perf_counter()
        if pipe is not None:
            pipe.run(steps2, gather)
        else:
            for _ in range(steps2):
                hp.step(gather)
        barrier()
        dt2 = time.perf_counter() - t1
        if use_dist:
            import torch
            t = torch.tensor([dt2], device="cpu" if args.emu else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        fixed_batch = {"value": round(npts * steps2 / args.steps / dt2 / 1e6, 3), "unit": "Mpoints/s", "ms_per_step": round(dt2 / steps2 * 1e3, 3), "selected_per_step": int(K_fixed),
                       "note": "the reference's reading: sampling()'s batch_size is ONE number for the round (here what one rank's 16 tiles select), the replicated chain does not grow with N"}
        set_batch(None)

    # ---- roofline leg (untimed): per-launch HIP-event timing of the instrumented kernels ------------------
    L = _lib.lib()

    def prof_rows(run):
        L.ssdr_prof_enable(1)
        run()
        rep = L.ssdr_prof_report().decode().strip().splitlines()
        L.ssdr_prof_enable(0)
        rows = []
        for ln in rep:
            name, calls, ms, work, work2 = ln.rsplit(" ", 4)
            rows.append((name, int(calls), float(ms), float(work), float(work2)))
        return sorted(rows, key=lambda r: -r[2])

    NPROF = 3
    def is_mfma(n_):          # the matrix-core kernel templates (per template since round 6; the 16 x 16 formulation keeps its two family names)
        return n_.startswith(("lfa32_", "lfa_att_kernel", "dense_bf16_kernel", "dense_chain_kernel", "dense_kernel"))
    mfma_peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_BF16_MFMA_TFLOPS
    mfma_insn = "v_mfma_f32_16x16x4_f32" if args.precision == "f32" else ("v_mfma_f32_16x16x32_bf16 (LocSE K = 10 and the d = 16 level on v_mfma_f32_16x16x4_f32)" if args.tiles16 else "v_mfma_f32_32x32x16_bf16 (every level; softmax over the neighbours inside the lane)")
    # (a) the way the timed region ran (every rank takes part because of the exchanges) ...
    timed_rows = [] if args.emu else prof_rows((lambda: (pipe.run(NPROF, gather), pipe.finish())) if pipe is not None else (lambda: [hp.step(gather) for _ in range(NPROF)]))
    roofline = None
    stage_ms = None
    stage_roofline = None
    whole_step = None
    if rank == 0 and not args.emu:
        # (b) ... and strictly sequential on rank 0: the kernel with the GPU to itself.  (b) is the roofline figure: it is
        # the kernel's own duration (rocprofv3's per-dispatch duration agrees with it, profiles/rNN_bench_seq_kernel_stats.csv),
        # whereas with several batches in flight an event pair on one stream also spans the time the dispatch waits
        # behind / shares the CUs with the other streams' kernels; (a) is reported beside it as "as_timed".
        hp.step(None)      # untimed: this state's first step grows its scratch buffers
        rows = prof_rows(lambda: [hp.step(None) for _ in range(NPROF)])
        # algorithmic work the library cannot know at enqueue time (counts that live on the device): the front end's one write of its rows
        # (28 B per voxel), the chamfer's point pairs (8 FLOP per pair and direction) from the last step's result
        sub_rows = int(hp.sub_m.to_host()[: hp.B].sum())
        T = hp._sel_static
        res = T["d_result"].to_host()
        n_unl = int(res[0])
        ref_sp = np.concatenate([res[8 + T["picks"]: 8 + T["picks"] + n_unl], np.array([s_ for b_ in range(hp.B) for s_ in hp.lab_rows.get(b_, [])], np.int64)]).astype(np.int64)
        sz = hp.sp_size_h[ref_sp].astype(np.float64); cl = hp.sp_cloud_h[ref_sp]
        pairs = float(sum(sz[cl == b_].sum() ** 2 - (sz[cl == b_] ** 2).sum() for b_ in range(hp.B)))      # ordered (source, target) pairs of points, i != j
        raw_pts = int(sum(len(r_[0]) for r_ in rooms))
        # fe_reduce is charged the operation's algorithmic bytes for the units its launch processes (28 B per input point, which it reads as
        # records, + 28 B per voxel, which it writes); the scatter the 28 B per point it reads
        extra_work = {"fe_reduce": 28.0 * (raw_pts + sub_rows) * NPROF, "sel_chamfer": 8.0 * pairs * NPROF}
        rows = [(r[0], r[1], r[2], r[3] + extra_work.get(r[0], 0.0), r[4]) for r in rows]
        # one-workgroup dependent chains occupy one CU of 256 and cost the pipelined step nothing (profiles/rNN_marginal.txt): they are listed with the
        # others, the roofline line is the longest CHIP-WIDE kernel
        one_cu = ("fps_chain",)
        # profiler sites that cover SEVERAL different kernels (latency chains of small launches): listed with the others, never the roofline line, which is
        # the single kernel TEMPLATE with the largest time per step (all launches of that template)
        multi_kernel_sites = ("knn_tree_handover", "fe_bbox_count_scan", "fe_rows_move", "tile_select", "knn_grid_build", "sel_candidate_rule", "sel_features_pack",
                              "sel_adjacency_propagate", "sel_clsbal", "sel_rank")
        f64_kernels = ("sel_chamfer",)
        name, calls, ms, work, work2 = [r for r in rows if r[0] not in one_cu and r[0] not in multi_kernel_sites][0]
        mfma = is_mfma(name)
        unit_div = 1e12 if mfma else 1e9
        achieved = work / (ms * 1e-3) / unit_div
        peak = mfma_peak if mfma else PEAK_HBM_GBS
        # HBM bytes per launch of that kernel from the committed PMC passes of this round (tools/collect_profiles.sh: rocprofv3 --pmc
        # FETCH_SIZE / --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md prescribes).
        # bench.py cannot collect PMC counters itself: the figure is a constant of that file, labelled with its source and commit.
        traffic, traffic_source, mfma_util, step_traffic = None, None, None, None
        try:
            pmcf = os.path.join("profiles", "%s_pmc_summary.json" % PROFILE_ROUND)
            pmc = json.load(open(os.path.join(ROOT, pmcf)))
            traffic = pmc["kernels"][name]["hbm_bytes_per_launch"]
            mfma_util = pmc["kernels"][name].get("mfma_utilisation")
            step_traffic = pmc.get("step_hbm_bytes")
            traffic_source = "%s (rocprofv3 --pmc passes at commit %s, not measured in this run)" % (pmcf, pmc.get("commit", "?"))
        except Exception:
            pass
        roofline = {"kernel": name, "bound": "mfma" if mfma else "hbm", "achieved": round(achieved, 3), "peak": peak,
                    "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "launches": calls, "avg_launch_us": round(ms * 1e3 / calls, 2), "conditions": "sequential pass, one kernel at a time",
                    "mfma_utilisation": mfma_util, "mfma_utilisation_is": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) of this kernel family in the "
                                                                          "committed PMC pass (same source as traffic)"}
        if mfma:
            roofline.update({"instruction": mfma_insn,
                             "achieved_is": "ALGORITHMIC FLOPs of the reference's formulation (attention dense d x d on every neighbour row) per second",
                             "executed_mfma": {"achieved": round(work2 / (ms * 1e-3) / 1e12, 3), "frac": round(work2 / (ms * 1e-3) / 1e12 / peak, 4),
                                               "note": "FLOPs the MFMA instructions execute: tile padding, both orientations of the position-encoding products, "
                                                       "three bf16 products per split product; the neighbour half of the attention product runs once per point in "
                                                       "dense_kernel (G rows) and is gathered, so it is not in this kernel"}})
        def family(r):
            d = {"ms_per_step": round(r[2] / NPROF, 3), "launches_per_step": r[1] // NPROF}
            rate = r[3] / (r[2] * 1e-3) if r[2] > 0 else 0.0
            if is_mfma(r[0]):
                d.update({"algorithmic_GFLOP_per_tile": round(r[3] / NPROF / TILES_PER_GPU / 1e9, 4), "algorithmic_TFLOPs": round(rate / 1e12, 2), "executed_mfma_TFLOPs": round(r[4] / (r[2] * 1e-3) / 1e12, 2), "frac_of_mfma_peak": round(rate / 1e12 / mfma_peak, 4)})
            elif r[0] in f64_kernels:
                d.update({"algorithmic_f64_TFLOPs": round(rate / 1e12, 2), "frac_of_f64_vector_peak": round(rate / 1e12 / PEAK_F64_VECTOR_TFLOPS, 4),
                          "note": "8 float64 FLOP per point pair of the reference's formulation; since round 5 the pairs are screened by v_mfma_f32_32x32x16_f16 "
                                  "(two half-precision pieces per coordinate, 11 of 16 k-slots) and only the winning run of four targets per source is evaluated "
                                  "in float64, so the float64 vector peak no longer bounds the kernel: it runs at the float32 vector issue rate "
                                  "(6 instructions per 4 x 64 pairs), csrc/select_chamfer.hip"})
            elif r[3] > 0:
                d.update({"algorithmic_GBs": round(rate / 1e9, 1), "frac_of_hbm_peak": round(rate / 1e9 / PEAK_HBM_GBS, 4)})
            if r[0] in one_cu:
                d["one_workgroup_chain"] = True
            return d
        roofline["others"] = {r[0]: family(r) for r in rows}
        for r in rows:
            if r[0] in multi_kernel_sites:
                roofline["others"][r[0]]["several_kernels"] = True
        # the two matrix-core families of rounds 1-5 as sums over their templates (continuity with the earlier rounds' roofline line)
        fam = {}
        for fname, pref in (("attention (lfa32_* templates)", ("lfa32_", "lfa_att_kernel")), ("per-point layers (dense_* templates)", ("dense_bf16_kernel", "dense_chain_kernel", "dense_kernel"))):
            rr = [r for r in rows if r[0].startswith(pref)]
            if rr:
                tms, w1, w2 = sum(r[2] for r in rr), sum(r[3] for r in rr), sum(r[4] for r in rr)
                fam[fname] = {"ms_per_step": round(tms / NPROF, 3), "launches_per_step": sum(r[1] for r in rr) // NPROF, "algorithmic_GFLOP_per_tile": round(w1 / NPROF / TILES_PER_GPU / 1e9, 3),
                              "algorithmic_TFLOPs": round(w1 / (tms * 1e-3) / 1e12, 2), "frac_of_mfma_peak": round(w1 / (tms * 1e-3) / 1e12 / mfma_peak, 4),
                              "executed_mfma_TFLOPs": round(w2 / (tms * 1e-3) / 1e12, 2)}
        roofline["families"] = fam
        for r in timed_rows:
            if r[0] == name and pipe is not None:
                roofline["as_timed"] = {"conditions": "%d batches in flight on %d streams" % (args.pipeline_depth, args.pipeline_depth),
                                        "achieved": round(r[3] / (r[2] * 1e-3) / unit_div, 3), "frac": round(r[3] / (r[2] * 1e-3) / unit_div / peak, 4),
                                        "avg_launch_us": round(r[2] * 1e3 / r[1], 2)}
        if world == 1:
            hp.step(None, timed_stages=True)
            stage_ms = {k: round(float(v), 3) for k, v in hp.timing.items()}
            # per-stage algorithmic work against the roofline that bounds the stage
            B, N0 = TILES_PER_GPU, ConfigS3DIS.num_points
            raw = int(sum(len(r[0]) for r in rooms)); sub = int(hp.sub_m.to_host()[:B].sum())
            stage_work = {"subsample+tile": ("hbm", 28.0 * (raw + sub) + 40.0 * B * N0), "knn_pyramid": ("hbm", 5.0e6 * B),
                          "randla_infer": ("mfma", 16.71e9 * B), "score": ("hbm", (52.0 + 8.0) * B * N0)}
            stage_roofline = {}
            for k, (bound, w) in stage_work.items():
                t = stage_ms[k] * 1e-3
                if bound == "hbm":
                    stage_roofline[k] = {"bound": "hbm", "algorithmic_bytes": int(w), "achieved_GBs": round(w / t / 1e9, 1), "frac": round(w / t / 1e9 / PEAK_HBM_GBS, 4)}
                else:
                    stage_roofline[k] = {"bound": "mfma", "algorithmic_flops": int(w), "achieved_TFLOPs": round(w / t / 1e12, 2), "peak_TFLOPs": mfma_peak,
                                         "frac": round(w / t / 1e12 / mfma_peak, 4), "hbm_bytes": int(63.1e6 * B), "hbm_frac": round(63.1e6 * B / t / 1e9 / PEAK_HBM_GBS, 4)}
            stage_roofline["select"] = {"bound": "latency", "note": "one-workgroup FPS chain (592 dependent picks) + chamfer graph (float64 values, screened on the matrix cores); overlapped with the other stages"}
            # the whole step against both rooflines: algorithmic bytes and FLOPs of all stages over the measured time per step
            alg_bytes = sum(w for b, w in stage_work.values() if b == "hbm") + 63.1e6 * B
            alg_flops = 16.71e9 * B
            tstep = dt / args.steps
            whole_step = {"ms_per_step": round(tstep * 1e3, 3), "algorithmic_bytes": int(alg_bytes), "algorithmic_flops": int(alg_flops),
                          "achieved_GBs": round(alg_bytes / tstep / 1e9, 1), "hbm_frac": round(alg_bytes / tstep / 1e9 / PEAK_HBM_GBS, 4),
                          "achieved_TFLOPs": round(alg_flops / tstep / 1e12, 2), "mfma_frac": round(alg_flops / tstep / 1e12 / mfma_peak, 4),
                          "measured_hbm_bytes_per_step": step_traffic,
                          "traffic_over_algorithmic": round(step_traffic / alg_bytes, 2) if step_traffic else None,
                          "traffic_source": traffic_source}
            if args.stages:
                print("stages(ms, sequential):", stage_ms, file=sys.stderr)

    # ---- one AL round at the reference's own scale (rank 0, N = 1; not the headline): inference over ALL rooms, then ONE selection ----
    # The headline step selects per 16-tile batch (the reference's picks-per-tile ratio), which under-represents the quadratic term of the
    # farthest-point chain 17-fold: the reference runs the network over all 272 rooms and then ONE GCN_FPS_sampling of batch_size = 10 000
    # over 2 x 10 000 candidates + the labelled rows (ssdr_main_S3DIS2.py:134, sampler2.py:736-781).  pipeline.ALRound is that round.
    al_round = None
    if rank == 0 and world == 1 and not args.emu and not args.no_al_round:
        try:
            ar = pipeline.ALRound(weights, rooms, args.al_batches, Cfg, batch_size=args.al_batch_size, precision=args.precision, selector=args.selector, tiles32=not args.tiles16)
            ar.run()                                   # untimed: grows the scratch buffers
            _lib.sync()
            t_al = []
            for _ in range(2):
                ta = time.perf_counter(); ar.infer_all()
                for st_ in ar.streams + ar.bstreams:
                    _lib.sync(st_)
                tb = time.perf_counter(); ar.sel._score_async(None); ar.sel._select_issue(None); sel_al, unl_al = ar.sel._select_collect(); tc = time.perf_counter()
                t_al.append((tc - ta, tb - ta, tc - tb))
            ta = time.perf_counter(); sel_al, unl_al = ar.run(); t_run = time.perf_counter() - ta          # the round as one enqueue sequence
            L.ssdr_prof_enable(1); ar.sel._score_async(None); ar.sel._select_issue(None); ar.sel._select_collect()
            rep = {ln.rsplit(" ", 4)[0]: float(ln.rsplit(" ", 4)[2]) for ln in L.ssdr_prof_report().decode().strip().splitlines()}
            L.ssdr_prof_enable(0)
            best = min(t_al)
            Tal = ar.sel._sel_static
            al_round = {"rooms": ar.tiles, "tile_points": int(ar.tile_points), "regions": int(ar.sel.S), "picks": int(len(sel_al)), "candidates": int(len(unl_al)), "labelled_rows": int(Tal["n_lab"]),
                        "ms": round(t_run * 1e3, 2), "Mpoints_per_s": round(ar.tile_points / t_run / 1e6, 2),
                        "inference_ms": round(best[1] * 1e3, 2), "selection_ms": round(best[2] * 1e3, 2), "fps_ms": round(rep.get("fps_chain", 0.0), 2),
                        "fps_us_per_pick": round(rep.get("fps_chain", 0.0) * 1e3 / max(len(sel_al), 1), 3), "chamfer_ms": round(rep.get("sel_chamfer", 0.0), 2),
                        "selection_families_ms": {k_: round(v_, 3) for k_, v_ in sorted(rep.items(), key=lambda kv: -kv[1])},
                        "selection_rule": ar.sel.rule_path,
                        "note": "front end -> KNN pyramid -> inference of %d batches of %d tiles (a batch per stream, four batches in flight), then scoring over all points and ONE "
                                "ssdr_gcn_fps_sampling_dev over all clouds' regions; ms = the whole round, GPU idle at both ends; inference_ms + selection_ms = the same with a "
                                "sync between the two halves; fps_ms = the farthest-point chain alone (hipEvent pair)" % (ar.nb, ar.B)}
            del ar
        except Exception as e:            # this leg is beside the headline: its failure must not cost the JSON line
            al_round = {"error": "%s: %s" % (type(e).__name__, e)}
            L.ssdr_prof_enable(0)

    # ---- CPU baseline leg (rank 0, N = 1 only): the oracle pipeline on ONE room/tile of the same workload ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pipeline_np
        cores = os.cpu_count() or 1
        ns = 4
        one = pipeline.HotPath(weights, ConfigS3DIS, precision=args.precision).load_rooms(rooms[:ns])
        tc = time.perf_counter()
        ref = pipeline_np.run(one, rooms[:ns], weights, threads=min(cores, ns))
        tcpu = time.perf_counter() - tc
        cpu = {"value": round(ns * ConfigS3DIS.num_points / tcpu / 1e6, 5), "unit": "Mpoints/s", "cores": cores, "kind": "port",
               "sample": "%d rooms / tiles of the same workload (%.1f s): C oracle for subsample (1 thread) + KNN (OpenMP over the %d tiles, as the "
                         "reference parallelises), NumPy with BLAS on all cores for RandLA-Net and selection" % (ns, tcpu, ns),
               "stage_ms": {k: round(float(v), 1) for k, v in ref["stage_ms"].items()}}
        # the two stages whose REAL reference code can run here (oracle/_ref, built from /root/reference in the build
        # container and shipped as a binary): same inputs, the reference's own parallel structure
        import oracle
        rlib = oracle.ref()
        if rlib is not None:
            tr = time.perf_counter()
            for r in rooms[:ns]:
                rlib.grid_subsampling(r[0], r[1].astype(np.float32), r[2].astype(np.int32), ConfigS3DIS.sub_grid_size)
            t_sub = time.perf_counter() - tr
            cur = ref["xyz"]
            tr = time.perf_counter()
            for ratio in ConfigS3DIS.sub_sampling_ratio:
                rlib.knn_batch(cur, cur, ConfigS3DIS.k_n, omp=True)
                nxt = cur[:, : cur.shape[1] // ratio]
                rlib.knn_batch(nxt, cur, 1, omp=True)
                cur = nxt
            t_knn = time.perf_counter() - tr
            cpu["reference_stage_ms"] = {"grid_subsampling (reference C++, 1 thread)": round(t_sub * 1e3, 1),
                                         "knn pyramid (reference C++, OpenMP over %d tiles)" % ns: round(t_knn * 1e3, 1)}

    if rank == 0:
        assert world == args.gpus or os.environ.get("SSDR_BENCH_FORCE_DIST"), "n_gpus must be what --gpus asked for"
        out = {"metric": METRIC, "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": DTYPE[args.precision], "data": "synthetic" if not args.emu else "synthetic (TEST MODE: CPU logic build, tiny workload - not a measurement)",
               "config": {"workload": "S3DIS-like rooms (synthetic, Area_5 seeds), %d rooms/tiles of 40960 points per GPU per step: grid-subsample "
                                      "dl=0.04 -> tile -> KNN pyramid k=16 [4,4,4,4,2] -> RandLA-Net infer (random-init weights, %s matrix products) -> WetSU/sb/clsbal "
                                      "ranking -> FPS-GCN select (gcn_number=1, gcn_top=0, %s)" % (tiles_per_gpu, args.precision, "FPS start fixed to candidate 0" if args.selector == "fps" else "global k-center over candidates + labelled regions"),
                          "tiles_per_gpu": tiles_per_gpu, "tile_points": Cfg.num_points, "raw_points_per_step_per_gpu": int(sum(len(r[0]) for r in rooms)),
                          "superpoints_per_gpu": int(hp.S), "selected_per_step": int(args.global_batch if args.global_batch > 0 else hp.select_per_tile * tiles_per_gpu * world), "sharding": "tiles",
                          "selection_batch": ("--global-batch %d: ONE batch_size for the job (the reference's reading)" % args.global_batch) if args.global_batch > 0 else
                                             ("%d regions per tile x the tiles of ALL ranks: the picks grow with N and the replicated global chain with N^2 (the harsher reading; "
                                              "`fixed_batch` is the same run with the reference's one batch_size per round)" % hp.select_per_tile),
                          "selection_rule": (getattr(pipe.hp[0], "rule_path", None) if pipe is not None else None) or getattr(hp, "rule_path", None),
                          **({"emulated_world": int(os.environ["SSDR_EMULATE_WORLD"])} if os.environ.get("SSDR_EMULATE_WORLD") else {}),
                          "batches_in_flight": args.pipeline_depth if pipe is not None else 1,
                          "timed_region": "GPU idle at both ends (every stream drained): K launch sequences of every stage and K completed selections, fill and "
                                          "drain of the %d-deep pipe included" % args.pipeline_depth if pipe is not None else "strictly sequential steps"},
               "stage_ms": stage_ms, "stage_roofline": stage_roofline, "whole_step": whole_step, "al_round": al_round, "fixed_batch": fixed_batch, "roofline": roofline, "cpu_baseline": cpu}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
