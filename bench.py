#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric  : inferences/sec at batch 256 (Model-A = embedding_47_krnl tables + FC 352-1024-512-256-1, fp32 FC),
          all tables resident in one GPU's HBM (BASELINE configs[1]); plus the embedding-gather HBM GB/s
          against peak on the largest model (Model-C, batch 4096) as the extra "gather" object.
step    : one pass of the hot path (index rows -> gather+pack -> 4-GEMM FC chain -> scores) over one batch
          of 256 synthetic requests whose index rows are already resident in HBM.
N > 1   : the path shards by independent request batches -> one replica per GPU, no data-path collective
          ("scaling": "weak"); value = batches all ranks processed / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3   # dense f32-input MFMA peak (same guide)
SEED_TABLES, SEED_IDX, SEED_WEIGHTS = 0xF1EE7, 1234, 99
N_IDX_BUFFERS = 32         # distinct index buffers rotated through, so caches are not re-hit artificially


def gather_bytes_per_inference(model, fr):
    """row reads + index reads + dense read + record write."""
    rows = sum(s.len * 4 for s in model.segments() if s.kind != fr.SEG_DENSE)
    idx = 4 * model.n_tables
    dense = 4 * model.dense_len
    write = 4 * model.record_len
    return rows + idx + dense + write


def pmc_field(kernel_prefix, field):
    """A per-kernel figure of the committed PMC summary (mfma_busy_fraction, mfma_f32_flops_per_launch, l2_hit_rate ...) or None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        for k, v in d.items():
            if k.startswith(kernel_prefix) and field in v:
                return v[field]
    except Exception:
        pass
    return None


def pmc_traffic(kernel_prefix):
    """HBM/fabric bytes per launch from the committed PMC summary (tools/pmc_traffic.sh: separate rocprofv3 --pmc passes over
    this same bench command, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md) -- or None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        for k, v in d.items():
            if k.startswith(kernel_prefix) and "traffic_bytes_per_launch" in v:
                return v["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def fc_flops_per_inference(fc):
    return 2 * sum(fc[i] * fc[i + 1] for i in range(4))


def main_sharded(args):
    """BASELINE configs[3]: Model-C, batch 4096, tables sharded by table-ID over the ranks; per step every rank gathers its
    [B x F] slice, ONE RCCL all-gather over xGMI rebuilds the records on every GPU, rank r runs the FC chain on its B/G items.
    value = B x steps / max-over-ranks time (one batch per step for the whole job: "scaling": "strong")."""
    import importlib
    import torch
    fr = graft.load_package()
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    env = dist_mod.DistEnv("nccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    G, r = env.world, env.rank
    B = 4096 if args.batch == 256 else args.batch
    model = fr.Model.builtin(fr.MODEL_C)
    ctx = fr.Context(model, device=env.local_rank, shard_rank=r, n_shards=G)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    offs, lens, F = model.shard_plan(G)
    dev = torch.device("cuda", env.local_rank)
    rng = np.random.default_rng(SEED_IDX)             # same request stream on every rank (replicated request)
    rows = model.rows()
    nbuf = 8
    idxs = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, model.n_tables)) * rows[None, :]).astype(np.int32)) for _ in range(nbuf)]
    dns = [fr.DeviceBuffer.from_numpy(ctx, rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32)) for _ in range(nbuf)]
    # Two workers and two sets of exchange buffers: while the FC chain of batch i-1 runs on worker B's stream, worker A gathers
    # batch i and the all-gather of batch i runs on torch's (RCCL) stream -- gather + exchange are hidden behind the MFMA work.
    wk = fr.Worker(ctx, B)       # gathers
    wk_fc = fr.Worker(ctx, B)    # FC chains
    a2a = args.exchange == "alltoall"
    if a2a and B % G:
        raise SystemExit("--exchange alltoall needs the batch divisible by the number of ranks")
    # slice transport: fp32, or the chain's own operand type (bf16: half the exchange bytes, e4m3: a quarter)
    lp = args.transport == "lp" and args.precision != "f32"
    prec_enum = {"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[args.precision]
    esz = {"f32": 4, "bf16": 2, "fp8": 1}[args.precision] if lp else 4
    local = [torch.empty((B, F * esz), dtype=torch.uint8, device=dev) for _ in range(2)]   # torch owns the exchange buffers (RCCL plumbing)
    gathered = [torch.empty((G, B // G if a2a else B, F * esz), dtype=torch.uint8, device=dev) for _ in range(2)]
    lo, hi = dist_mod.item_range(r, G, B)
    scores = torch.empty((max(hi - lo, 1),), dtype=torch.float32, device=dev)

    def exchange(k):
        if a2a:
            env.all_to_all_slices(local[k], gathered[k])
        else:
            env.all_gather_slices(local[k], gathered[k])

    def fc(k):
        tp = prec_enum if lp else fr.FC_FP32
        if a2a:
            wk_fc.fc_from_slices_lp(B // G, 0, hi - lo, gathered[k].data_ptr(), tp, scores.data_ptr())
        else:
            wk_fc.fc_from_slices_lp(B, lo, hi - lo, gathered[k].data_ptr(), tp, scores.data_ptr())

    # Stream-ordered hand-over, no host synchronisation inside a step: the gather worker's stream, torch's exchange stream and the FC
    # worker's stream wait for one another through events (torch.cuda.ExternalStream wraps the workers' hipStream_t).
    s_gather = torch.cuda.ExternalStream(wk.stream_ptr(), device=dev)
    s_fc = torch.cuda.ExternalStream(wk_fc.stream_ptr(), device=dev)
    s_x = torch.cuda.Stream(device=dev)            # the exchange (RCCL) stream
    state = {"pending": None, "fc_done": [None, None]}   # pending: buffer set gathered + exchanged, waiting for its FC chain

    def step(i):
        k = i & 1
        if state["fc_done"][k] is not None:
            s_gather.wait_event(state["fc_done"][k])   # the FC chain that read gathered[k] / local[k] two steps ago has finished
        wk.gather_slices(B, idxs[i % nbuf], dns[i % nbuf], local[k].data_ptr(), prec_enum if lp else fr.FC_FP32)   # async, gather worker's stream
        if state["pending"] is not None:
            p = state["pending"]
            s_fc.wait_stream(s_x)                      # its exchange has landed
            fc(p)                                      # async on the FC worker's stream: overlaps this step's gather + exchange
            ev = torch.cuda.Event()
            ev.record(s_fc)
            state["fc_done"][p] = ev
        s_x.wait_stream(s_gather)                      # the slice must be complete before RCCL reads it
        with torch.cuda.stream(s_x):
            exchange(k)
        state["pending"] = k

    def drain():
        if state["pending"] is not None:
            s_fc.wait_stream(s_x)
            fc(state["pending"])
            state["pending"] = None
        wk.sync()
        wk_fc.sync()
        torch.cuda.synchronize()
        state["fc_done"] = [None, None]

    if args.precision != "f32":   # the slices travel as fp32; the FC chain re-packs them to bf16 / e4m3 operands
        ctx.set_fc_precision(fr.FC_BF16 if args.precision == "bf16" else fr.FC_FP8)
        if args.precision == "fp8":   # activation exponents from the first batch's gathered slices (same on every rank)
            cal_l = torch.empty((B, F), dtype=torch.float32, device=dev)   # calibration always sees fp32 slices
            cal_g = torch.empty((G, B // G if a2a else B, F), dtype=torch.float32, device=dev)
            wk.gather_only(B, idxs[0], dns[0], cal_l.data_ptr())
            wk.sync()
            if a2a:
                env.all_to_all_slices(cal_l, cal_g)
            else:
                env.all_gather_slices(cal_l, cal_g)
            torch.cuda.synchronize()
            if a2a:
                wk_fc.calibrate_fp8_slices(B // G, 0, hi - lo, cal_g.data_ptr())
            else:
                wk_fc.calibrate_fp8_slices(B, 0, B, cal_g.data_ptr())
            del cal_l, cal_g
    for i in range(args.warmup):
        step(i)
    drain()
    env.barrier(); torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()                                            # the last batch's FC chain is inside the timed region
    env.barrier(); torch.cuda.synchronize(); ctx.synchronize()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    # outside the timed region: the last pipelined batch against the same batch run step by step with host synchronisation
    piped = scores.clone()
    last = args.steps - 1
    wk.gather_slices(B, idxs[last % nbuf], dns[last % nbuf], local[0].data_ptr(), prec_enum if lp else fr.FC_FP32)
    wk.sync()
    exchange(0)
    torch.cuda.synchronize()
    fc(0)
    wk_fc.sync()
    torch.cuda.synchronize()
    verified = bool(torch.equal(piped, scores)) if args.steps > 0 else None
    if r == 0:
        print(json.dumps({
            "metric": "inferences/sec, Model-C batch 4096, tables sharded by table-ID", "value": B * args.steps / dt, "unit": "inferences/s",
            "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "Model-C batch=%d, %d-way table-ID shards (slice F=%d floats), 1 all-gather of [B x F] per step, "
                                   "FC on B/G items per rank" % (B, G, F), "parallelism": "table-sharded x%d" % G, "exchange": args.exchange,
                       "slice_transport": args.precision if lp else "f32", "pipelined_equals_stepwise": verified,
                       "exchange_bytes_in_per_rank_per_step": int(G * (B // G if a2a else B) * F * esz)}}))
    wk.close()
    wk_fc.close()
    ctx.close()
    env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40000)
    ap.add_argument("--warmup", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=2, help="host driver threads (reference THREAD_NUM = 4, constant.h:42)")
    ap.add_argument("--depth", type=int, default=2, help="workers (streams) each driver thread keeps in flight")
    ap.add_argument("--sweep", action="store_true", help="also print a threads x depth sweep to stderr (experiments)")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="replicas: BASELINE configs[1] (default, the headline metric); sharded: Model-C batch 4096 with tables "
                         "sharded by table-ID over the ranks + one RCCL all-gather of the looked-up slices (BASELINE configs[3])")
    ap.add_argument("--model", choices=["A", "B", "C"], default="A", help="A = BASELINE configs[1] (default headline); B/C: other configs")
    ap.add_argument("--precision", choices=["f32", "bf16", "fp8"], default="f32",
                    help="FC chain arithmetic (bf16 = BASELINE configs[2], fp8 = configs[4]: e4m3, calibrated on the first batch)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the multi-stream throughput loop and the Model-C leg: only the single-stream roofline launches run, so "
                         "that a rocprofv3 --kernel-trace --stats summary of this command shows the kernel under the roofline's conditions")
    ap.add_argument("--transport", choices=["f32", "lp"], default="lp",
                    help="sharded mode: slices travel as fp32, or (lp) in the chain's own operand type when --precision is bf16 / fp8")
    ap.add_argument("--exchange", choices=["allgather", "alltoall"], default="allgather",
                    help="sharded mode: all-gather every slice to every rank (BASELINE configs[3]) or all-to-all only each rank's items")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true", help="plumbing test: ranks share the visible GPU(s) (use with --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-zipf", action="store_true", help="skip the zipf-index repeat of the Model-C gather leg (PMC passes: one index law per kernel name)")
    ap.add_argument("--no-model-c", action="store_true", help="skip the Model-C batch-4096 gather roofline leg")
    args = ap.parse_args()

    if args.mode == "sharded":
        return main_sharded(args)
    import importlib
    import torch
    fr = graft.load_package()
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    env = dist_mod.DistEnv(args.backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    rank, world = env.rank, env.world
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n_dev = max(fr.device_count(), 1)
    local_rank = env.local_rank % n_dev if args.share_device else env.local_rank  # --share-device: plumbing test on one GPU
    dist = env.dist

    fr = graft.load_package()
    if fr.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")

    B = args.batch
    model = fr.Model.builtin({"A": fr.MODEL_A, "B": fr.MODEL_B, "C": fr.MODEL_C}[args.model])
    ctx = fr.Context(model, device=local_rank)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    if args.precision == "bf16":
        ctx.set_fc_precision(fr.FC_BF16)
    rng = np.random.default_rng(dist_mod.replica_seed(SEED_IDX, rank))
    rows = model.rows()
    n_bufs = N_IDX_BUFFERS if args.model == "A" else 8
    idx_host = [(rng.random((B, model.n_tables)) * rows[None, :]).astype(np.int32) for _ in range(n_bufs)]
    d_idx = [fr.DeviceBuffer.from_numpy(ctx, a) for a in idx_host]
    d_dense = ([fr.DeviceBuffer.from_numpy(ctx, rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32)) for _ in range(n_bufs)]
               if model.dense_len else None)
    if args.precision == "fp8":
        ctx.set_fc_precision(fr.FC_FP8)
        dense_host = rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32) if model.dense_len else None
        cal = fr.Worker(ctx, B)
        cal.calibrate_fp8(idx_host[0], dense_host)   # activation exponents from one batch of the same index law
        cal.close()
    if args.model != "A" or args.precision != "f32":
        # non-headline configurations: throughput line only
        driver = fr.Driver(ctx, args.threads, args.depth, B)
        driver.run_resident(B, args.warmup, d_idx, d_dense)
        ctx.synchronize()
        el = driver.run_resident(B, args.steps, d_idx, d_dense)
        print(json.dumps({"metric": "inferences/sec", "value": args.steps * B / el, "unit": "inferences/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "dtype": args.precision,
                          "data": "synthetic", "config": {"workload": "Model-%s batch=%d %s FC chain, index rows resident in HBM" % (
                              args.model, B, args.precision), "fc_tflops": fc_flops_per_inference(model.fc) * B * args.steps / el / 1e12}}))
        driver.close()
        ctx.close()
        return
    driver = fr.Driver(ctx, args.threads, args.depth, B)

    def barrier():
        env.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ctx.synchronize()

    if args.sweep and rank == 0:
        for th, dp in ((1, 1), (1, 2), (1, 4), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2), (8, 4), (16, 2)):
            dv = fr.Driver(ctx, th, dp, B)
            dv.run_resident(B, 200, d_idx)
            el = dv.run_resident(B, 2000, d_idx)
            print("sweep threads=%d depth=%d: %.2f us/batch, %.2f M inf/s" % (th, dp, 1e6 * el / 2000, 2000 * B / el / 1e6), file=sys.stderr)
            dv.close()

    if args.roofline_only:
        args.steps, args.warmup, args.no_cpu_baseline, args.no_model_c = 8, 8, True, True
    driver.run_resident(B, args.warmup, d_idx)
    barrier()
    t0 = time.perf_counter()
    driver.run_resident(B, args.steps, d_idx)
    barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)   # MAX over ranks

    result = None
    if rank == 0:
        value = world * args.steps * B / dt
        result = {
            "metric": "inferences/sec at batch 256", "value": value, "unit": "inferences/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Model-A (embedding_47_krnl: 47 tables, 1.415 GB) batch=%d, fp32 FC 352-1024-512-256-1, "
                                   "all tables resident in one GPU's HBM; hash-filled tables, uniform indices, "
                                   "index rows resident in HBM" % B,
                       "batch": B, "driver_threads": args.threads, "workers_per_thread": args.depth,
                       "parallelism": "replicas x%d" % world},
        }
        # ---- roofline of the dominant kernel: measured live with HIP events on the worker's stream ----
        # Model-A streams through the fused item-tile kernel: ONE launch = the whole hot path (gather + 4 GEMMs) of `group` queued
        # batches, 64 items per workgroup at the default group of 64 (32 items for smaller groups).  (Models that do not fit
        # LDS use fr_pipeline_kernel<-1>, group = 1.)
        group = ctx.stream_group()
        wk = fr.Worker(ctx, B)
        ring = [fr.DeviceBuffer(ctx, B * 4) for _ in range(max(8, 2 * group))]
        t_warm = time.time()   # >= 1 s of back-to-back launches first: the shader clock ramps over many milliseconds of load, and
        while time.time() - t_warm < 1.0:   # with --roofline-only nothing else has loaded the chip before this point
            for i in range(4 * group):
                wk.push_device(B, d_idx[i % N_IDX_BUFFERS], None, ring[i % len(ring)])
            wk.sync()
        launches = 200 if group > 1 else 1000
        if group == 1:
            for i in range(8):   # refill the stage pipeline so that every timed launch carries all five stages
                wk.push_device(B, d_idx[i % N_IDX_BUFFERS], None, ring[i % len(ring)])
        wk.timer_start()
        for i in range(launches * group):
            wk.push_device(B, d_idx[i % N_IDX_BUFFERS], None, ring[i % len(ring)])
        pipe_ms = wk.timer_stop_ms() / launches
        wk.sync()
        fc = model.fc
        flops = fc_flops_per_inference(fc) * B * group
        ach = flops / (pipe_ms * 1e-3) / 1e12
        wpe = int(os.environ.get("FR_FUSED_WPE", "4"))  # workgroups are built for 4 waves per SIMD (two per CU) unless forced to 2
        if group >= 64 and os.environ.get("FR_FUSED_M2", "1") != "0":
            kname = "fr_fused_tile_m2_kernel<44>"   # 64 items per workgroup: one launch of 64 batches covers the 256 CUs
        elif group > 1:
            kname = "fr_fused_tile_kernel<2, 44, %d, %s>" % (wpe, "false" if wpe == 4 else "true")
        else:
            kname = "fr_pipeline_kernel<-1, 0>"
        result["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                              "frac": ach / MFMA_F32_PEAK_TF, "traffic": pmc_traffic(kname.split(",")[0] if group > 1 else kname),
                              "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; "
                                                "FETCH_SIZE x2 gfx950 correction), bytes per launch",
                              "kernel": "%s: one launch = the whole hot path (gather + the 4-GEMM chain) of %d queued batches of %d, "
                                        "back-to-back on ONE stream" % (kname, group, B),
                              "batches_per_launch": group, "avg_launch_ms": pipe_ms, "algorithmic_flops_per_launch": flops,
                              "pmc_mfma_busy_fraction": pmc_field(kname.split(",")[0] if group > 1 else kname, "mfma_busy_fraction"),
                              "pmc_mfma_f32_flops_per_launch": pmc_field(kname.split(",")[0] if group > 1 else kname, "mfma_f32_flops_per_launch"),
                              "note": "`value` above runs %d such streams concurrently" % (args.threads * args.depth)}
        # per-stage launches (unpipelined submit path), for reference
        d_sc = ring[0]
        wk.submit_device(B, d_idx[0], None, d_sc)
        wk.sync()
        rec = wk.records_dptr()
        layer_ms = []
        for layer in range(4):
            for _ in range(20):
                wk.fc_layer_only(B, layer)
            wk.sync()
            wk.timer_start()
            for _ in range(200):
                wk.fc_layer_only(B, layer)
            layer_ms.append(wk.timer_stop_ms() / 200)
            wk.sync()
        result["roofline"]["single_stage_avg_launch_ms"] = layer_ms
        reps = 300
        # gather kernel at the bench batch
        for _ in range(20):
            wk.gather_only(B, d_idx[0], None, rec)
        wk.sync()
        wk.timer_start()
        for i in range(reps):
            wk.gather_only(B, d_idx[i % N_IDX_BUFFERS], None, rec)
        g_ms = wk.timer_stop_ms() / reps
        wk.sync()
        wk.close()
        for b_ in ring:
            b_.free()
        gb = gather_bytes_per_inference(model, fr) * B
        result["gather_roofline_bench_batch"] = {"bound": "hbm", "achieved": gb / (g_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": gb / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "avg_launch_ms": g_ms, "algorithmic_bytes_per_launch": gb}

        # ---- PCIe-inclusive rate (host index buffers through fr_worker_submit/sync; reported, never `value`) ----------
        if not args.roofline_only and world == 1:
            hd = fr.Driver(ctx, args.threads, 4, B)
            hd.run_host(B, 200, idx_host)
            el = hd.run_host(B, 2000, idx_host)
            hd.close()
            result["pcie_inclusive"] = {"value": 2000 * B / el, "unit": "inferences/s", "ms_per_step": 1e3 * el / 2000,
                                        "what": "index rows start in host memory: memcpy to pinned -> H2D -> 5 stage launches -> D2H -> sync per batch, "
                                                "%d threads x 4 workers" % args.threads}
            # the same host-resident stream through fr_worker_push_host: pinned staging blocks, one H2D + one fused launch + one D2H
            # per block of batches, scores delivered to host memory -- still PCIe-inclusive, still never `value`
            hs = fr.Driver(ctx, args.threads, args.depth, B)
            hs.run_host(B, 2000, idx_host, streaming=True)
            n_s = 20000
            el = hs.run_host(B, n_s, idx_host, streaming=True)
            hs.close()
            result["pcie_inclusive_streaming"] = {"value": n_s * B / el, "unit": "inferences/s", "ms_per_step": 1e3 * el / n_s,
                                                  "what": "index rows start in host memory, scores end in host memory: blocks of batches staged in pinned "
                                                          "memory, one H2D + one fused launch + one D2H per block, %d threads x %d workers"
                                                          % (args.threads, args.depth)}
        # ---- CPU baseline: the oracle ("port": C, OpenMP over items) on this node's host cores, bounded sample (~10 s).
        #      (The 4-GEMM chain through numpy/OpenBLAS sgemm was measured 3x SLOWER than the oracle's own loops at this
        #      batch size on the 128-core host -- threading overhead on 256-row matrices -- so the oracle's chain is used.)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (bench contract)
            O = graft.load_oracle()
            om = O.OracleModel("A")
            ws = [ctx.get_weights(l) for l in range(4)]
            nthreads = O.lib().oracle_num_threads()
            n_done, t_cpu, t_g = 0, 0.0, 0.0
            t_start = time.perf_counter()
            while time.perf_counter() - t_start < 10.0:
                a = idx_host[n_done % N_IDX_BUFFERS]
                t1 = time.perf_counter()
                r = om.gather(a, content_mode=O.FILL_HASH, seed=SEED_TABLES)
                t2 = time.perf_counter()
                om.fc_chain(r.view(np.float32), ws, acc64=False)
                t3 = time.perf_counter()
                t_cpu += t3 - t1
                t_g += t2 - t1
                n_done += 1
            result["cpu_baseline"] = {"value": n_done * B / t_cpu, "unit": "inferences/s", "cores": nthreads, "kind": "port",
                                      "sample": "%d batches of %d (Model-A, same seeded indices): oracle C port -- OpenMP gather with "
                                                "on-the-fly hash tables (%.0f%% of the time) + fp32 4-GEMM chain" % (n_done, B, 100.0 * t_g / t_cpu)}

    driver.close()
    for b in d_idx:
        b.free()
    ctx.close()

    # ---- the headline gather measurement: Model-C, batch 4096 (rank 0, N=1 only) -------------------------
    if rank == 0 and world == 1 and not args.no_model_c:
        try:
            mc = fr.Model.builtin(fr.MODEL_C)
            cc = fr.Context(mc, device=local_rank)
            cc.fill_tables(fr.FILL_HASH, SEED_TABLES)
            BC = 4096
            rng = np.random.default_rng(SEED_IDX)
            rows = mc.rows()
            nbuf = 8
            idxs = [fr.DeviceBuffer.from_numpy(cc, (rng.random((BC, mc.n_tables)) * rows[None, :]).astype(np.int32)) for _ in range(nbuf)]
            dns = [fr.DeviceBuffer.from_numpy(cc, rng.uniform(-1, 1, (BC, mc.dense_len)).astype(np.float32)) for _ in range(nbuf)]
            wk = fr.Worker(cc, BC)
            rec = wk.records_dptr()
            for i in range(10):
                wk.gather_only(BC, idxs[i % nbuf], dns[i % nbuf], rec)
            wk.sync()
            reps = 100
            wk.timer_start()
            for i in range(reps):
                wk.gather_only(BC, idxs[i % nbuf], dns[i % nbuf], rec)
            ms = wk.timer_stop_ms() / reps
            wk.sync()
            gb = gather_bytes_per_inference(mc, fr) * BC
            result["gather"] = {"workload": "Model-C (2x embedding_377_krnl + 64 dense: 376 tables, 63.2 GB) batch=4096, uniform indices",
                                "bound": "hbm", "achieved": gb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": gb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("gather_pack_xcd_kernel<8>"),
                                "traffic_source": "profiles/r01_pmc_traffic.json (PMC, FETCH_SIZE x2 correction), bytes per launch", "avg_launch_ms": ms,
                                "algorithmic_bytes_per_launch": gb, "inferences_per_s": BC / (ms * 1e-3),
                                "kernel": "gather_pack_xcd_kernel<8>"}
            # same kernel, zipf(1.05) indices per table (SURVEY 8d input (ii)): popular rows repeat inside a batch, the repeats
            # are merged by the texture-address coalescer / served by L2, so no explicit ballot/shuffle dedup pass is needed
            def zipf_idx(shape, rows, alpha=1.05):
                u = rng.random(shape)
                x = ((rows[None, :].astype(np.float64) ** (1.0 - alpha) - 1.0) * u + 1.0) ** (1.0 / (1.0 - alpha))
                return np.minimum(np.floor(x) - 1, rows[None, :] - 1).astype(np.int32)
            zs = [zipf_idx((BC, mc.n_tables), rows) for _ in range(0 if args.no_zipf else nbuf)]
            if zs:
                dup = float(np.mean([1.0 - len(np.unique(z[:, t])) / BC for z in zs[:2] for t in range(0, mc.n_tables, 7)]))
                zidx = [fr.DeviceBuffer.from_numpy(cc, z) for z in zs]
                for i in range(10):
                    wk.gather_only(BC, zidx[i % nbuf], dns[i % nbuf], rec)
                wk.sync()
                wk.timer_start()
                for i in range(reps):
                    wk.gather_only(BC, zidx[i % nbuf], dns[i % nbuf], rec)
                zms = wk.timer_stop_ms() / reps
                wk.sync()
                result["gather"]["zipf_1.05"] = {"avg_launch_ms": zms, "achieved": gb / (zms * 1e-3) / 1e9, "frac": gb / (zms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "duplicate_fraction_within_batch": dup}
            wk.close()
            cc.close()
        except Exception as ex:  # the main metric must still be reported
            result["gather"] = {"error": str(ex)}

    if rank == 0:
        print(json.dumps(result))
    env.close()


if __name__ == "__main__":
    main()
