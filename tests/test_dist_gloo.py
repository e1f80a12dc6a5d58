"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: the replica mode's timing rule and the table-sharded mode's
ONE exchange step (all-gather of [B x F] slices) + item partition.  The oracle stands in for the per-rank kernels here;
the same orchestration runs on RCCL in bench.py --mode sharded."""
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
from conftest import free_port_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json, time
    import numpy as np
    sys.path.insert(0, %(root)r)
    import __graft_entry__ as g
    fr = g.load_package()
    O = g.load_oracle()
    import importlib
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    env = dist_mod.DistEnv("gloo")
    assert env.world == 2 and env.backend == "gloo"
    import torch
    # --- replica mode: barrier + max-over-ranks timing, whole-job aggregation --------------------------------------
    env.barrier()
    mine = 0.5 + 0.25 * env.rank
    mx = env.max_over_ranks(mine)
    assert abs(mx - 0.75) < 1e-12, mx
    assert dist_mod.replica_seed(1234, 0) != dist_mod.replica_seed(1234, 1)
    assert dist_mod.aggregate_throughput(1000, mx, env.world) == 2 * 1000 / 0.75
    assert abs(env.sum_over_ranks(env.rank + 1) - 3.0) < 1e-12
    # fp8 chain: every rank adopts the element-wise MIN of the per-rank activation exponents
    assert env.min_over_ranks_int([3 + env.rank, 5 - env.rank, -2, 7]) == [3, 4, -2, 7]
    # --- sharded mode: every rank sees the whole request, gathers only its slice, one all-gather --------------------
    which, name = %(which)d, %(name)r
    model = fr.Model.builtin(which)
    om = O.OracleModel(name)
    offs, lens, F = model.shard_plan(env.world)
    B = 24
    rng = np.random.default_rng(99)                      # same seed on both ranks = replicated request
    rows = model.rows()
    idx = (rng.random((B, model.n_tables)) * np.minimum(rows, 5000)[None, :]).astype(np.int32)
    dense = rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32) if model.dense_len else None
    full = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=5)      # [B][K] uint32
    local = np.zeros((B, F), np.uint32)
    local[:, :lens[env.rank]] = full[:, offs[env.rank]:offs[env.rank] + lens[env.rank]]   # what gather_only writes on this shard
    gathered = env.all_gather_slices(torch.from_numpy(local.view(np.int32))).numpy().view(np.uint32)
    assert gathered.shape == (2, B, F)
    rec = dist_mod.assemble_records(gathered, offs, lens, model.record_len)
    assert np.array_equal(rec, full)
    lo, hi = dist_mod.item_range(env.rank, env.world, B)
    ws = [np.full(om.fc[i] * om.fc[i + 1], 1.0 / om.fc[i], np.float32) for i in range(4)]
    mine_scores = om.fc_chain(rec[lo:hi].view(np.float32), ws)
    pad = np.zeros(B, np.float32); pad[lo:hi] = mine_scores
    tot = torch.from_numpy(pad.copy()); env.dist.all_reduce(tot)
    ref = om.fc_chain(full.view(np.float32), ws)
    assert np.array_equal(tot.numpy(), ref)
    # --- all-to-all variant: every rank receives only ITS items' slices (1/world of the all-gather traffic) --------
    mine = env.all_to_all_slices(torch.from_numpy(local.view(np.int32))).numpy().view(np.uint32)
    assert mine.shape == (2, B // 2, F)
    assert (lo, hi) == (env.rank * B // 2, (env.rank + 1) * B // 2)
    assert np.array_equal(dist_mod.assemble_records(mine, offs, lens, model.record_len), full[lo:hi])
    cover = [dist_mod.item_range(r, 2, B) for r in range(2)]
    assert cover[0][0] == 0 and cover[0][1] == cover[1][0] and cover[1][1] == B
    env.close()
    print("rank %%d ok" %% env.rank)
''')


@pytest.mark.parametrize("which,name", [(0, "A"), (2, "C")])
def test_two_rank_gloo(tmp_path, which, name, fr, O):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "which": which, "name": name})
    port = free_port_block(1)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out)
        assert "rank %d ok" % r in out


def test_shard_plan_properties(fr):
    for which in (fr.MODEL_A, fr.MODEL_B, fr.MODEL_C):
        m = fr.Model.builtin(which)
        segs = m.segments()
        starts = {s.rec_offset for s in segs}
        for G in (1, 2, 3, 4, 8):
            offs, lens, F = m.shard_plan(G)
            assert offs[0] == 0 and sum(lens) == m.record_len and F == max(lens) and F % 4 == 0
            for g in range(G):
                assert offs[g] in starts                      # slices are whole segments
                if g:
                    assert offs[g] == offs[g - 1] + lens[g - 1]
            assert max(lens) - min(lens) <= 64                # float-balanced (largest row is 32 floats, dense block 64)
    with pytest.raises(fr.FleetRecError):
        fr.Model.builtin(fr.MODEL_A).shard_plan(48)           # more shards than segments


def test_sharding_only_when_the_tables_do_not_fit(fr):
    """north_star: "tables shard by table-ID across the 8 GPUs of one node ... only when total table bytes exceed one GPU's 288 GB".
    Model.min_shards applies that rule on the host: the three reference models need no sharding, Model-C with its tables inflated 5 x
    (322 GB: BASELINE configs[4]) needs 2 shards (configs[4] uses all 8), and a model with one table larger than a GPU cannot be served by table-ID sharding."""
    for which in (fr.MODEL_A, fr.MODEL_B, fr.MODEL_C):
        m = fr.Model.builtin(which)
        assert m.min_shards() == 1
        assert sum(m.shard_table_bytes(1)) == m.table_bytes()
        assert sum(m.shard_table_bytes(8)) >= m.table_bytes()      # a COPY pad may pull a table into a second shard
    big = fr.Model.builtin(fr.MODEL_C).clone(row_scale=5.0)       # BASELINE configs[4]: 322 GB of tables
    assert big.table_bytes() > 288e9
    G = big.min_shards()
    assert G == 2 and max(big.shard_table_bytes(2)) < 0.9 * 288e9 < max(big.shard_table_bytes(1))   # 159 + 163 GB: two GPUs hold it
    assert max(big.shard_table_bytes(8)) < max(big.shard_table_bytes(2))     # configs[4] runs it 8-way; the rule's minimum is 2
    bigger = fr.Model.builtin(fr.MODEL_C).clone(row_scale=9.0)
    assert bigger.min_shards() in (4, 8) and max(bigger.shard_table_bytes(bigger.min_shards())) <= 0.9 * 288e9
    assert max(bigger.shard_table_bytes(bigger.min_shards() // 2)) > 0.9 * 288e9
    huge = fr.Model.builtin(fr.MODEL_C).clone(row_scale=40.0)     # one table alone outgrows a GPU: table-ID sharding cannot split it
    assert huge.min_shards() is None


def test_single_rank_exchange_fills_the_callers_buffer(fr):
    """world == 1: the exchange is the identity, but a caller-supplied output buffer must still receive the slice (bench.py's
    sharded mode reads ITS buffer; round 1 left it uninitialised -- ADVICE r01)."""
    import importlib
    import torch
    import __graft_entry__ as g
    g.load_package()
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    env_keep = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        env = dist_mod.DistEnv(None)
        assert env.world == 1 and env.dist is None
        local = torch.arange(24, dtype=torch.uint8).reshape(4, 6)
        for fn in (env.all_gather_slices, env.all_to_all_slices):
            out = torch.full((1, 4, 6), 255, dtype=torch.uint8)
            ret = fn(local, out)
            assert ret is out and torch.equal(out[0], local)
            assert tuple(fn(local).shape) == (1, 4, 6)
        assert env.min_over_ranks_int([4, -1]) == [4, -1]
    finally:
        for k, v in env_keep.items():
            if v is not None:
                os.environ[k] = v


def _run_bench(args, timeout=240, detail=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    if detail:
        env["FR_BENCH_DETAIL"] = detail   # where bench.py writes the full result (default: gpurun_out/bench_detail.json)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _the_line(out):
    """bench.py's stdout contract: the LAST line is ONE compact JSON object the driver can parse -- under 4 KB, with
    the contract keys, `roofline` and `cpu_baseline`; nothing else on stdout starts with '{'."""
    import json
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.rstrip().splitlines()[-1] == lines[0], out[-3000:]
    assert len(lines[0]) < 4096, len(lines[0])
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j, k
    assert "workload" in j["config"]
    return j


def test_gemm_workgroups_from_the_kernel_name():
    """bench.py sizes the side-by-side roofline measurement of a stage-pipeline row from the kernel name the library reports: workgroups per
    launch -> how many launches fit on the chip's 256 compute units together."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_wg_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.gemm_workgroups("fc_pp_gemm_kernel<1, 3>", 2048, 4096) == 128                      # Model-C FC1, 256 x 256 tiles: two launches fit
    assert b.gemm_workgroups("fc_lp_gemm_kernel<1, 2, 256, 2, 8, 16>", 2048, 4096) == 128      # ... the same tile's plain loop
    assert b.gemm_workgroups("fc_lp_gemm_kernel<0, 2, 128, 2, 8, 32>", 2048, 4096) == 256      # fp32, 128 x 256 tiles: the chip
    assert b.gemm_workgroups("fc_gemm_pipe_kernel<1, 5, 1>", 512, 4096) == 64                   # FC2 on 128 x 256 tiles
    assert b.gemm_workgroups("fc_pp_gemm_n128_kernel<1, 2>", 2048, 4096) == 256                 # a lone worker's FC1: 128 x 256 tiles cover the chip
    assert b.gemm_workgroups("fc_lp_gemm_kernel<2, 1, 128, 2, 8, 32>", 256, 4096) == 64         # FC3 on 128 x 128 tiles
    assert b.gemm_workgroups("fr_pipeline_kernel<4, 1>", 1, 4096) is None and b.gemm_workgroups(None, 1, 1) is None


def test_compact_line_of_a_full_result_stays_under_4k():
    """The writer of the stdout line, fed the 21.7 KB result of round 3's driver command (the one the driver could not parse): the line it
    makes of it is < 4 KB, carries `roofline` (bound / achieved / peak / unit / frac / traffic) and `cpu_baseline` (value / unit / cores /
    kind / sample), and points at the detail file.  A result padded with 40 more configuration rows still fits (optional summaries drop)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_line_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r03_bench_line_driver_cmd.json")))
    assert len(json.dumps(full)) > 20000
    for i, c in enumerate(full["configs"]):
        c["tag"] = "row%d_%s" % (i, c["dtype"])
    s = bench.compact_line(full)
    j = json.loads(s)
    assert len(s) < 4096 and "\n" not in s
    assert all(k in j["roofline"] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_name", "avg_launch_ms"))
    assert all(k in j["cpu_baseline"] for k in ("value", "unit", "cores", "kind", "sample"))
    assert j["value"] == pytest.approx(full["value"], rel=1e-5) and j["gather_per_bank"]["frac"] == pytest.approx(full["gather_per_bank"]["frac"], rel=1e-5)
    assert len(j["other_configs"]) == len(full["configs"]) and j["detail"].endswith("bench_detail.json")
    full["configs"] = full["configs"] * 5
    for i, c in enumerate(full["configs"]):
        full["configs"][i] = dict(c, tag="a_rather_long_configuration_tag_%03d" % i)
    s = bench.compact_line(full)
    assert len(s) < 4096 and "roofline" in json.loads(s) and "cpu_baseline" in json.loads(s)


def test_bench_self_launches_ranks_and_refuses_a_wrong_world():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start 2 rank processes itself (before any GPU call) and report
    n_gpus == 2; a WORLD_SIZE that contradicts --gpus is refused instead of being reported as a different n_gpus.
    --plumbing-only = launcher + gloo rendezvous + barrier + max-over-ranks timing, no GPU and no library."""
    import json
    rc, out, err = _run_bench(["--gpus", "2", "--plumbing-only", "--steps", "7", "--warmup", "3"])
    assert rc == 0, err
    j = _the_line(out)                               # ONE compact JSON line, from rank 0
    assert "roofline" in j and "cpu_baseline" in j   # (empty objects in this mode: the line writer is the real run's)
    assert j["n_gpus"] == 2 and j["steps"] == 7 and j["warmup"] == 3 and j["config"]["self_launched"] is True
    assert j["ms_per_step"] * 7 >= 20.0 - 1e-6       # max over ranks: rank 1 sleeps 20 ms
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--plumbing-only"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"refusing" in p.stderr and b"{" not in p.stdout


def test_bench_launcher_fails_fast_when_a_rank_dies():
    """a rank that dies before a barrier must not leave its peers in the collective until the backend's timeout.
    `--fail-rank 1` makes rank 1 exit with status 3 before the rendezvous; the launcher sees it, stops the other ranks and exits
    non-zero within seconds (it used to wait for the ranks one by one)."""
    import time
    t0 = time.time()
    rc, out, err = _run_bench(["--gpus", "3", "--plumbing-only", "--fail-rank", "1"], timeout=120)
    dt = time.time() - t0
    assert rc == 3, (rc, err[-2000:])
    assert dt < 30.0, dt
    assert "rank 1 exited with status 3" in err and "{" not in out


def test_bench_failed_collective_leg_keeps_the_line_and_exits_non_zero(tmp_path):
    """VERDICT r03 item 6 / ADVICE r03: a sharded leg that fails on one rank after the headline was measured.  The line (with
    `sharded_error`) must still be the last stdout line and the job's exit status must be non-zero -- whichever rank failed: the failing
    rank leaves with status 4, the launcher stops its peers, and rank 0 -- possibly inside a collective's C code when the SIGTERM
    arrives -- prints the line from its signal-watch thread before it leaves."""
    import time
    for bad in (1, 0):
        t0 = time.time()
        rc, out, err = _run_bench(["--gpus", "2", "--plumbing-only", "--fail-sharded-rank", str(bad)], timeout=120, detail=str(tmp_path / "d.json"))
        assert rc == 4, (rc, err[-2000:])
        assert time.time() - t0 < 60.0
        j = _the_line(out)
        assert j["n_gpus"] == 2 and "sharded_error" in j and "roofline" in j, j
        assert "injected failure" in err


def test_bench_eight_ranks_rehearsal_on_the_cpu_back_end(tmp_path):
    """the default `bench.py --gpus 8` line with EIGHT ranks before an 8-GPU node ever sees it.  The test boxes have one
    GPU and their process guard allows at most six processes on it, so the eight ranks run on the library's CPU back-end (--device -1:
    fr_ctx_create(device = -1) on every rank, gloo, row-capped tables, token step counts): the same control flow -- self-launch, rendezvous,
    barriers and max-over-ranks timing of the headline, the voted per-rank legs (gather_per_bank_all_ranks, configs_all_ranks), the two
    table-sharded legs under their LineGuard with Model-C batch 4096 cut EIGHT ways (slice plan 496 / 496 / 504 / 488 / 496 ..., F = 504,
    512 items per rank, one all-gather of [4096 x 504] slices per step) and configs[4]'s inflated row counts -- so that the first real
    8-GPU run cannot die on plan arithmetic, a missing key or a hung collective.  Asserts every leg's keys, rank 0's scores of its 512 items
    bit-identical to an unsharded CPU context, and the printed wall-time budget."""
    import json
    import time
    detail = str(tmp_path / "detail.json")
    t0 = time.time()
    slow_build = "san.so" in os.path.basename(os.environ.get("FR_LIB", ""))   # tools/run_sanitizers.sh: instrumented host code, eight ranks of it on eight cores
    rc, out, err = _run_bench(["--gpus", "8", "--device", "-1", "--backend", "gloo", "--rows-cap", "2000", "--steps", "4", "--warmup", "2"], timeout=1800 if slow_build else 600, detail=detail)
    wall = time.time() - t0
    assert rc == 0, err[-3000:]
    line = _the_line(out)
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["scaling"] == "weak" and "REHEARSAL" in line["data"]
    assert line["gather_per_bank_all_ranks"]["ranks_measured"] == 8 and len(line["configs_all_ranks"]) == 3
    assert all(c["ranks"] == 8 and c["inf_per_s"] > 0 for c in line["configs_all_ranks"].values())
    for k in ("sharded", "sharded_inflated_fp8"):
        assert line[k]["n_gpus"] == 8 and line[k]["value"] > 0 and line[k]["scaling"] == "strong" and line[k]["exchange"] == "allgather", line[k]
    # "ok" = the leg's own verification.  The rehearsal has no pipelined form: the sharded leg compares rank 0's scores with an unsharded
    # CPU context's bit for bit (True); the row-inflated leg compares nothing and says so (None, not a constant True: ADVICE r05)
    assert line["sharded"]["ok"] is True and line["sharded_inflated_fp8"].get("ok") is None, (line["sharded"], line["sharded_inflated_fp8"])
    assert "sharded_error" not in line and "roofline" in line and "cpu_baseline" in line
    j = json.load(open(detail))
    for k in ("headline", "gather_per_bank_all_ranks", "configs_all_ranks", "sharded", "sharded_inflated_fp8"):
        assert j["leg_seconds"][k] > 0, j["leg_seconds"]
    assert abs(sum(j["leg_seconds"].values()) - j["leg_seconds_total"]) < 0.2 and j["leg_seconds_total"] <= wall
    m_ = re.search(r"bench.py: wall time per leg \(s\): .*; total ([0-9.]+) s", err)
    assert m_ and (float(m_.group(1)) < 300.0 or slow_build), err[-1500:]
    c3, c4 = j["sharded"]["config"], j["sharded_inflated_fp8"]["config"]
    assert c3["slice_lens"] == [496, 496, 504, 488, 496, 496, 496, 496] and sum(c3["slice_lens"]) == 3968 and c3["items_this_rank"] == [0, 512]
    assert c3["exchange_bytes_in_per_rank_per_step"] == 8 * 4096 * 504 * 4 and c3["sharded_vs_unsharded_context"]["bit_identical"] is True
    assert j["sharded"]["baseline_config"] == "configs[3]" and j["sharded_inflated_fp8"]["baseline_config"] == "configs[4]" and "rows x 5" in c4["workload"]


def test_bench_eight_rank_sharded_alltoall_rehearsal():
    """`bench.py --mode sharded --exchange alltoall` with eight ranks on the CPU back-end: each rank receives only ITS 512 items' slices."""
    import json
    rc, out, err = _run_bench(["--gpus", "8", "--device", "-1", "--backend", "gloo", "--mode", "sharded", "--exchange", "alltoall", "--rows-cap", "2000",
                               "--steps", "2", "--warmup", "1"], timeout=600)
    assert rc == 0, err[-3000:]
    j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    c = j["config"]
    assert j["n_gpus"] == 8 and c["exchange"] == "alltoall" and c["exchange_bytes_in_per_rank_per_step"] == 8 * 512 * 504 * 4
    assert c["sharded_vs_unsharded_context"]["bit_identical"] is True


@pytest.mark.gpu
def test_bench_default_line_is_compact(gpu, tmp_path):
    """`bench.py --gpus 1 --quick` (every leg of the driver's command, short timed regions): the stdout line parses, is < 4 KB and carries
    roofline + cpu_baseline; the full result is in the detail file."""
    import json
    detail = str(tmp_path / "detail.json")
    rc, out, err = _run_bench(["--gpus", "1", "--quick", "--steps", "20", "--warmup", "5"], timeout=1500, detail=detail)
    assert rc == 0, err[-3000:]
    j = _the_line(out)
    rf, cb = j["roofline"], j["cpu_baseline"]
    assert rf["bound"] == "mfma" and 0.3 < rf["frac"] <= 1.0 and rf["unit"] == "TFLOP/s" and rf["avg_launch_ms"] > 0 and "traffic" in rf, rf
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"], cb
    assert j["value"] > 0 and j["value_pcie_inclusive"] > 0 and j["config"]["batches_per_launch"] >= 1
    assert 0.3 < j["gather_per_bank"]["frac"] <= 1.0
    full = json.load(open(detail))
    assert len(full["configs"]) >= 8 and all("roofline" in c or "error" in c for c in full["configs"])
    assert len(j["other_configs"]) == sum(1 for c in full["configs"] if c.get("value"))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--mode", "sharded", "--rows-cap", "20000", "--steps", "3", "--warmup", "1"],
                                   ["--mode", "sharded", "--rows-cap", "20000", "--steps", "3", "--warmup", "1", "--precision", "fp8", "--exchange", "alltoall"],
                                   ["--mode", "sharded", "--rows-cap", "20000", "--row-scale", "1.5", "--steps", "3", "--warmup", "1", "--precision", "bf16"]])
def test_bench_two_ranks_on_one_gpu(gpu, extra):
    """The N = 2 path end to end on a one-GPU box: self-launched ranks, gloo rendezvous, both ranks on device 0
    (--share-device).  Replicas: value aggregates both ranks.  Sharded: 2-way table-ID shards, the exchange through gloo, scores
    of rank 0's items equal to an unsharded context (1e-5 in fp32, bit-identical in fp8) and the pipelined run equal to the stepwise one."""
    import json
    # the DEFAULT line (extra == []) keeps its roofline leg: at N > 1 rank 0 prices the dominant kernel on its own replica
    args = ["--gpus", "2", "--backend", "gloo", "--share-device", "--legs", "none" if extra else "roofline"] + (extra or ["--steps", "300", "--warmup", "100"])
    detail = None if extra else os.path.join(os.environ.get("TMPDIR", "/tmp"), "fr_bench_detail_%d.json" % os.getpid())
    rc, out, err = _run_bench(args, timeout=1200, detail=detail)
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0
    if not extra:   # the default line is the compact one; the objects checked below live in the detail file
        line = _the_line(out)
        assert line["sharded"]["value"] > 0 and line["gather_per_bank_all_ranks"]["ranks_measured"] == 2 and len(line["configs_all_ranks"]) == 3, line
        j = json.load(open(detail))
        os.unlink(detail)
    if extra:
        c = j["config"]
        assert c["pipelined_equals_stepwise"] is True
        chk = c["sharded_vs_unsharded_context"]
        if "--row-scale" in extra:   # BASELINE configs[4]'s switch: the inflated model is not expected to fit one GPU, no unsharded twin
            assert chk is None and "rows x 1.5" in c["workload"] and c["shard_table_bytes_this_rank"] > 0
        elif j["dtype"] == "f32":   # B/G items per rank vs batch B unsharded: a different split-K plan, sums differ in the last bits
            assert chk["max_rel_err"] <= 1e-5, c
        else:                     # low-precision chains sum whole K per output in one order: bit for bit
            assert chk["bit_identical"] is True, c
    else:
        assert j["timed_batches"] >= 300 and j["timed_s"] >= 2.0 and j["scaling"] == "weak"
        g = j["gather_per_bank_all_ranks"]     # N > 1: every rank gathers on its own Model-C replica, rank 0 reports the sum
        assert g["ranks_measured"] == 2 and g["achieved"] > 0 and g["peak"] == 16000.0, g
        ca = j["configs_all_ranks"]            # ... and Model-C bf16 / fp8 and Model-B bf16 on every rank's replica
        assert [c_["dtype"] for c_ in ca] == ["bf16", "fp8", "bf16"] and all(c_["ranks_measured"] == 2 and c_["value"] > 0 for c_ in ca), ca
        # the driver's own command measures the table-sharded split (BASELINE configs[3]): Model-C batch 4096, 2-way
        # table-ID shards, one all-gather per step, bf16 transport -- on the default line, against an unsharded context
        sh = j["sharded"]
        assert sh["value"] > 0 and sh["n_gpus"] == 2 and sh["dtype"] == "bf16" and sh["scaling"] == "strong" and sh["baseline_config"] == "configs[3]", sh
        c = sh["config"]
        # (B / G items per rank take other GEMM tiles than the unsharded batch of 4096 -- 16x16x32 vs 32x32x16 MFMAs sum the k of an
        # instruction in another order -- so a bf16 rounding of an activation may flip: the bf16 chain's own 5e-3, not bit identity)
        assert c["pipelined_equals_stepwise"] is True and c["sharded_vs_unsharded_context"]["max_rel_err"] <= 5e-3 and c["slice_transport"] == "bf16", c
        assert "sharded_inflated_fp8" not in j      # 316 GB do not fit the one GPU both ranks share here
        rf = j["roofline"]                          # the N > 1 line carries the roofline object too
        assert rf["bound"] == "mfma" and 0.3 < rf["frac"] <= 1.0 and rf["avg_launch_ms"] > 0, rf


@pytest.mark.gpu
def test_bench_four_ranks_on_one_gpu(gpu, tmp_path):
    """The default N > 1 line with as many ranks as a one-GPU box allows on its card (the pool's process guard: six processes, one of them
    this test runner): four self-launched ranks on device 0 (--share-device, gloo), row-capped replicas so that four of them fit, every leg of
    the line -- replicas headline, rank 0's roofline leg, the per-rank gather and configuration legs, the 4-way table-sharded step (bf16
    transport) against an unsharded context -- and the printed wall-time budget.  (Eight ranks: the CPU-back-end rehearsal above.)"""
    import json
    detail = str(tmp_path / "detail.json")
    rc, out, err = _run_bench(["--gpus", "4", "--backend", "gloo", "--share-device", "--rows-cap", "20000", "--legs", "roofline", "--steps", "300", "--warmup", "100", "--quick"],
                              timeout=1200, detail=detail)
    assert rc == 0, err[-3000:]
    line = _the_line(out)
    assert line["n_gpus"] == 4 and line["value"] > 0 and line["roofline"]["bound"] == "mfma" and 0.3 < line["roofline"]["frac"] <= 1.0
    assert line["gather_per_bank_all_ranks"]["ranks_measured"] == 4 and len(line["configs_all_ranks"]) == 3
    assert line["sharded"]["n_gpus"] == 4 and line["sharded"]["ok"] is True and line["sharded"]["dtype"] == "bf16"
    j = json.load(open(detail))
    assert j["sharded"]["config"]["sharded_vs_unsharded_context"]["max_rel_err"] <= 5e-3
    m_ = re.search(r"bench.py: wall time per leg \(s\): .*; total ([0-9.]+) s", err)
    assert m_ and float(m_.group(1)) < 600.0, err[-1500:]


@pytest.mark.gpu
def test_bench_rccl_code_path_on_one_rank(gpu, tmp_path):
    """The one-GPU boxes cannot run two RCCL ranks (RCCL refuses two ranks on one device), so the N > 1 line's RCCL code path had never executed
    anywhere.  `--force-process-group --backend nccl` builds a ONE-rank torch.distributed process group on RCCL and sends every collective of
    the line through it: max / sum over ranks on device tensors, the voted per-rank legs, the sharded leg's all-gather on RCCL's stream with
    the external-stream hand-over between the gather worker, the exchange and the FC worker.  Also guards the bench contract's stdout: librccl
    prints a five-line version banner to STDOUT when its first communicator comes up (DistEnv points fd 1 at stderr meanwhile) -- the line
    must be the ONLY thing on stdout."""
    import json
    detail = str(tmp_path / "detail.json")
    rc, out, err = _run_bench(["--gpus", "1", "--backend", "nccl", "--force-process-group", "--legs", "none", "--quick", "--rows-cap", "200000"], timeout=900, detail=detail)
    assert rc == 0, err[-3000:]
    assert len(out.strip().splitlines()) == 1, out[:2000]          # no RCCL banner in front of the line
    line = _the_line(out)
    assert line["n_gpus"] == 1 and line["gather_per_bank_all_ranks"]["ranks_measured"] == 1 and len(line["configs_all_ranks"]) == 3
    assert line["sharded"]["ok"] is True and line["sharded"]["dtype"] == "bf16"
    j = json.load(open(detail))
    assert j["sharded"]["config"]["sharded_vs_unsharded_context"]["bit_identical"] is True     # one shard = the whole record: the same kernels
    for prec, ex in (("fp8", "alltoall"), ("bf16", "allgather")):
        rc, out, err = _run_bench(["--gpus", "1", "--backend", "nccl", "--force-process-group", "--mode", "sharded", "--precision", prec, "--exchange", ex,
                                   "--rows-cap", "200000", "--steps", "10", "--warmup", "3"], timeout=600)
        assert rc == 0, err[-3000:]
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and len(out.strip().splitlines()) == 1, out[:1000]
        c = json.loads(lines[0])["config"]
        assert c["pipelined_equals_stepwise"] is True and c["sharded_vs_unsharded_context"]["bit_identical"] is True and c["exchange"] == ex, c


@pytest.mark.gpu
def test_bench_two_ranks_other_configuration(gpu):
    """`bench.py --gpus 2 --model B --batch 1024 --precision bf16`: the non-headline configurations aggregate over the ranks too (same number
    of batches on every rank, barriers on both sides, slowest rank's time)."""
    import json
    rc, out, err = _run_bench(["--gpus", "2", "--backend", "gloo", "--share-device", "--model", "B", "--batch", "1024", "--precision", "bf16", "--quick"], timeout=900)
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["dtype"] == "bf16" and j["timed_batches"] > 0
    assert abs(j["value"] - 2 * j["timed_batches"] * 1024 / j["timed_s"]) <= 1e-6 * j["value"]


def test_bench_tcp_leg_fails_cleanly_without_a_gpu():
    """bench.py's tcp leg (fleetrec_sender -> fleetrec_server --stream in their own processes) on a host without a GPU: the server's
    set-up fails loudly (no CPU fallback), the leg reports it and leaves no process behind instead of hanging."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base = bench.free_port_block(4)
    assert 20000 <= base < 60000
    import __graft_entry__ as g
    if g.load_package().device_count() > 0:
        pytest.skip("a GPU is visible: the GPU suite covers the working path (tests/test_gpu_server.py)")
    t0 = time.time()
    r = bench.leg_tcp(256, 0, threads=2, total=64)
    assert ("error" in r or "skipped" in r) and "value" not in r, r
    assert time.time() - t0 < 60


@pytest.mark.gpu
def test_bench_cabi_rccl_leg_child_and_failure_path(gpu, tmp_path):
    """Round 6: at N > 1 the bench line also runs the C-ABI's OWN table-sharded step over RCCL (fr_comm_init_rank + fr_worker_submit_sharded with
    G = N ranks) -- the one piece of the product a one-GPU box cannot execute with G > 1 -- each rank in a CHILD process, so that an RCCL set-up
    that fails or hangs is a string in `sharded_cabi_rccl`, never a lost line.  Here: (1) the child itself with a one-rank communicator whose unique
    id is drawn by this (living) process, as the bench parent does: fp32 within 1e-5 of the unsharded context, bf16 steps timed; (2) the parent's
    integration with two ranks SHARING the one GPU (FR_BENCH_CABI_RCCL=force): RCCL refuses two ranks on one device, the line carries that
    error, `ranks_ok` 0, and the job still exits 0 with its headline and its torch.distributed sharded leg intact."""
    import json
    import __graft_entry__ as g
    fr = g.load_package()
    out = str(tmp_path / "child.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cabi-child", "0/1/%d/%s/%s" % (gpu, fr.Comm.unique_id().hex(), out)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    res = json.load(open(out))
    assert res["ok"] is True and res["world"] == 1 and res["fp32_max_rel_err_vs_unsharded_first_512"] <= 1e-5 and res["inferences_per_s"] > 1e5, res
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["FR_BENCH_CABI_RCCL"] = "force"
    env["FR_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--legs", "none", "--steps", "300", "--warmup", "100"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = _the_line(p.stdout.decode())
    c = line["sharded_cabi_rccl"]
    assert c["ok"] is False and c["ranks_ok"] == 0 and c["world"] == 2 and "ncclCommInitRank" in c["error"], c
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["sharded"]["ok"] is True
