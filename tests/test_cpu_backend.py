"""The CPU back-end behind the same symbols (fr_ctx_create(..., device = -1); SURVEY section 8(b), BASELINE configs[0]): the library's
own host code (csrc/fr_cpu.cpp) against the oracle.  Runs without a GPU.  Bar: records bit-exact; fp32 scores within 2e-6 of the
fp64-accumulating oracle (max-abs over max|ref|, as in tests/test_gpu_*.py: a k-ordered fp32 chain over K = 3968 terms carries ~1e-6
of rounding by itself -- the oracle's own fp32 chain sits at 0.8e-6 from its fp64 one); the reference's known answers exact."""
import ctypes
import os
import re
import subprocess
import time

import numpy as np
import pytest
from conftest import free_port_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "gpu-fpga-recommendation-system_amd", "host")
SEED_TABLES, SEED_WEIGHTS = 0xF1EE7, 99
NAMES = {0: "A", 1: "B", 2: "C"}
CPU = -1


def rel_err(got, ref):
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def uniform_idx(rng, rows, B):
    return (rng.random((B, len(rows))) * rows[None, :]).astype(np.int32)


@pytest.mark.parametrize("which", [0, 1, 2])
def test_records_bit_exact_and_scores_vs_oracle(fr, O, which):
    """Models A / B / C (rows capped at 20 000 so that the tables fit any host), hashed tables, random weights, uniform per-table indices:
    gather_only bit-exact, submit + sync and fc_only within 2e-6 of the fp64-accumulating oracle, the two bit-identical to each other,
    ragged batches bit-identical to the same items inside a larger batch (the CPU chain has one summation order whatever the batch)."""
    m = fr.Model.builtin(which).clone(max_rows=20000)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=CPU)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    rng = np.random.default_rng(11 + which)
    B = 300
    idx = uniform_idx(rng, m.rows(), B)
    idx[0], idx[1] = 0, m.rows() - 1
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    wk = fr.Worker(ctx, 512)
    rec = wk.gather_records(idx, dense).reshape(B, m.record_len)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    assert np.array_equal(rec, want)
    scores = wk.infer(idx, dense)
    ws = [ctx.get_weights(l) for l in range(4)]
    ref = om.fc_chain(want.view(np.float32), ws, acc64=True)
    assert rel_err(scores, ref) <= 2e-6, rel_err(scores, ref)
    assert np.array_equal(wk.fc_scores(want.view(np.float32)), scores)
    for b in (1, 2, 3, 5, 63):
        assert np.array_equal(wk.infer(idx[:b], None if dense is None else dense[:b]), scores[:b]), b
    # the resident-buffer entry points (what fr_driver_run_resident calls): "device" memory of a CPU context is host memory
    d_i = fr.DeviceBuffer.from_numpy(ctx, idx)
    d_d = fr.DeviceBuffer.from_numpy(ctx, dense) if dense is not None else None
    d_s = fr.DeviceBuffer(ctx, B * 4)
    wk.submit_device(B, d_i, d_d, d_s)
    wk.sync()
    assert np.array_equal(d_s.download(np.float32, B), scores)
    wk.push_device(B, d_i, d_d, d_s)
    wk.sync()
    assert np.array_equal(d_s.download(np.float32, B), scores)
    # the fills are the device's fills (csrc/fr_content.h is one definition for both back-ends) = the oracle's content function
    t = m.n_tables // 2
    assert np.array_equal(ctx.download_table(t, 7, 5), rec_rows(O, om, m, t, 7, 5))
    wk.close()
    ctx.close()


def rec_rows(O, om, m, t, row0, n):
    """Rows [row0, row0 + n) of table t as the oracle's content function fills them (through a one-column gather)."""
    idx = np.zeros((n, m.n_tables), np.int32)
    idx[:, t] = np.arange(row0, row0 + n)
    dense = np.zeros((n, m.dense_len), np.float32) if m.dense_len else None
    full = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    seg = [s for s in m.segments() if s.kind == 0 and s.src == t][0]
    return full[:, seg.rec_offset:seg.rec_offset + seg.len]


@pytest.mark.parametrize("which", [0, 1, 2])
def test_reference_literal_mode_known_answer(fr, O, which):
    """The reference's own run on the CPU back-end: even/odd tables (host.cpp:66-88), ONE index per item broadcast to every table, the 32
    fixed indices (embedding_47_krnl.cpp:899-914), all-ones weights (cuda_server.c:152-160) -> K * H1 * H2 * H3 or 0, exactly."""
    m = fr.Model.builtin(which).clone(max_rows=200, index_mode=fr.INDEX_PER_ITEM)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=CPU)
    ctx.fill_tables(fr.FILL_EVEN_ODD, 0)
    ctx.fill_weights(fr.WEIGHTS_ONES, 0)
    idx = np.tile(om.halves[0].idx_random, 3)
    B = len(idx)
    dense = np.tile(np.where(idx % 2 == 0, 1.0, 0.0).astype(np.float32)[:, None], (1, m.dense_len)) if m.dense_len else None
    wk = fr.Worker(ctx, B)
    rec = wk.gather_records(idx[:, None], dense).reshape(B, m.record_len)
    assert np.array_equal(rec, om.gather(idx, dense=dense, content_mode=O.FILL_EVEN_ODD))
    assert all((rec[j] == (0x3F800000 if idx[j] % 2 == 0 else 0)).all() for j in range(B))
    fc = m.fc
    val = np.float32(float(fc[0]) * fc[1] * fc[2] * fc[3])
    assert np.array_equal(wk.infer(idx[:, None], dense), np.where(idx % 2 == 0, val, np.float32(0)))
    wk.close()
    ctx.close()


def test_readme_known_answers(fr):
    """GPU/final_network_cublasLt_1_node_no_FIFO_scatter/README.md:7-11: K = 512 -> 2^36, K = 1024 -> 2^37 (batch 128, constant.h:32)."""
    for K, want in ((512, 2.0 ** 36), (1024, 2.0 ** 37)):
        T = fr.TableDesc(mem_class=0, table_id=0, source=0, dim=K, rows=4, bank=0, round=0, addr_axi=0)
        S = fr.Segment(kind=fr.SEG_TABLE, src=0, src_col=0, rec_offset=0, len=K, source=0)
        d = fr.ModelDesc()
        d.name = b"readme"
        d.n_tables, d.n_segments = 1, 1
        d.tables = ctypes.pointer(T)
        d.segments = ctypes.pointer(S)
        d.record_len, d.dense_len = K, 0
        for i, v in enumerate((K, 1024, 512, 256, 1)):
            d.fc[i] = v
        m = fr.Model(ctypes.pointer(d), keepalive=(T, S, d))
        ctx = fr.Context(m, device=CPU)
        ctx.upload_table(0, np.ones((4, K), np.float32))
        ctx.fill_weights(fr.WEIGHTS_ONES, 0)
        wk = fr.Worker(ctx, 128)
        assert (wk.infer(np.zeros((128, 1), np.int32)) == np.float32(want)).all()
        wk.close()
        ctx.close()


@pytest.mark.parametrize("mode", ["bank", "blocked"])
def test_bank_contract_and_3node_buffer(fr, O, mode):
    """FR_INDEX_PER_BANK (one index per memory bank, bank-interleaved rows: the kernel's contract, embedding_377_krnl.cpp:1261-1290) and
    FR_LAYOUT_BLOCKED (the 3-node server's receive buffer, 3-node cuda_server.c:515,541,566) on the CPU back-end, Model-C."""
    base = fr.Model.builtin(fr.MODEL_C)
    om = O.OracleModel("C")
    rng = np.random.default_rng(5)
    B = 96
    if mode == "bank":
        m = base.clone(max_rows=5000, index_mode=fr.INDEX_PER_BANK)
        idx = uniform_idx(rng, m.index_ranges(), B)
    else:
        m = base.clone(max_rows=5000, layout=fr.LAYOUT_BLOCKED)
        idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    ctx = fr.Context(m, device=CPU)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    wk = fr.Worker(ctx, B)
    rec = wk.gather_records(idx, dense)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES, per_bank=(mode == "bank"))
    ws = [ctx.get_weights(l) for l in range(4)]
    if mode == "bank":
        assert np.array_equal(rec.reshape(B, -1), want)
        x = want.view(np.float32)
    else:   # [CPU: B x 64][FPGA0: B x 1952][FPGA1: B x 1952], then read by the GEMM as B x 3968 item-major
        blocked = np.concatenate([want[:, 0:64].ravel(), want[:, 64:2016].ravel(), want[:, 2016:3968].ravel()])
        assert np.array_equal(rec, blocked)
        x = blocked.view(np.float32).reshape(B, m.record_len)
    ref = om.fc_chain(x, ws, acc64=True)
    assert rel_err(wk.infer(idx, dense), ref) <= 2e-6
    wk.close()
    ctx.close()


def test_errors_and_what_the_cpu_back_end_refuses(fr):
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=1000)
    ctx = fr.Context(m, device=CPU)
    wk = fr.Worker(ctx, 64)
    idx = np.zeros((8, m.n_tables), np.int32)
    with pytest.raises(fr.FleetRecError) as e:          # tables not filled
        wk.infer(idx)
    assert e.value.status == fr.FR_ERR_STATE
    ctx.fill_tables(fr.FILL_HASH, 1)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    good = wk.infer(idx)
    bad = idx.copy()
    bad[3, 5] = m.rows()[5]                             # one past the end of table 5 (reference: silent OOB, embedding_47_krnl.cpp:927-933)
    with pytest.raises(fr.FleetRecError) as e:
        wk.infer(bad)
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    assert np.array_equal(wk.infer(idx), good)          # the flag does not stick
    wk.idx[:8] = idx
    wk.submit(8)
    with pytest.raises(fr.FleetRecError) as e:          # one batch in flight per worker, as on the device
        wk.submit(8)
    assert e.value.status == fr.FR_ERR_STATE
    wk.sync()
    for prec in (fr.FC_BF16, fr.FC_FP8):                # fp32 only
        with pytest.raises(fr.FleetRecError) as e:
            ctx.set_fc_precision(prec)
        assert e.value.status == fr.FR_ERR_INVALID and "fp32 only" in str(e.value)
    ctx.set_fc_precision(fr.FC_FP32)
    sc = np.empty(8, np.float32)
    with pytest.raises(fr.FleetRecError) as e:          # host-fed streaming is a device path
        wk.push_host(idx, None, sc)
    assert e.value.status == fr.FR_ERR_STATE and "CPU back-end" in str(e.value)
    with pytest.raises(fr.FleetRecError):
        wk.fc_layer_only(8, 0)
    with pytest.raises(fr.FleetRecError):
        wk.calibrate_fp8(idx)
    with pytest.raises(fr.FleetRecError):
        fr.Comm.init_rank(ctx, b"\0" * 128)
    assert fr.cpu_set_threads(2) == 2 and np.array_equal(wk.infer(idx), good)   # the thread count changes no bit
    assert fr.cpu_set_threads(0) >= 1
    with pytest.raises(fr.FleetRecError):
        fr.cpu_set_threads(-3)
    wk.close()
    ctx.close()
    # a device >= 0 never falls back to the CPU
    if fr.device_count() == 0:
        with pytest.raises(fr.FleetRecError) as e:
            fr.Context(m, device=0)
        assert e.value.status == fr.FR_ERR_NO_DEVICE
    # tables that cannot fit the host are refused up front instead of being killed half-way through the fill
    huge = fr.Model.builtin(fr.MODEL_C).clone(row_scale=40.0)
    with pytest.raises(fr.FleetRecError) as e:
        fr.Context(huge, device=CPU)
    assert e.value.status == fr.FR_ERR_OOM


def test_table_sharded_contexts_on_the_cpu(fr, O):
    """BASELINE configs[3]'s data flow through the CPU back-end: Model-C's eight table-ID shards as eight CPU contexts, every shard's slice
    bit-exact, and fr_worker_fc_from_slices on the all-gathered layout for every rank's B/G items = the unsharded context's scores, bit
    for bit (the chain's summation order does not depend on how the record was assembled)."""
    import importlib
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    G, B = 8, 100
    m = fr.Model.builtin(fr.MODEL_C).clone(max_rows=3000)
    om = O.OracleModel("C")
    offs, lens, F = m.shard_plan(G)
    rng = np.random.default_rng(8)
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    full = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    shards, slices = [], []
    for r in range(G):
        c = fr.Context(m, device=CPU, shard_rank=r, n_shards=G)
        c.fill_tables(fr.FILL_HASH, SEED_TABLES)
        c.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
        wk = fr.Worker(c, B)
        sl = wk.gather_records(idx, dense).reshape(B, F)
        assert np.array_equal(sl[:, :lens[r]], full[:, offs[r]:offs[r] + lens[r]]), r
        with pytest.raises(fr.FleetRecError):
            wk.infer(idx, dense)                         # submit on a sharded context: as on the device, use the sharded entry points
        shards.append((c, wk))
        slices.append(sl)
    gathered = np.stack(slices)
    whole = fr.Context(m, device=CPU)
    whole.fill_tables(fr.FILL_HASH, SEED_TABLES)
    whole.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    w0 = fr.Worker(whole, B)
    ref = w0.infer(idx, dense)
    scores = np.empty(B, np.float32)
    for r, (c, wk) in enumerate(shards):
        lo, hi = dist_mod.item_range(r, G, B)
        d_g = fr.DeviceBuffer.from_numpy(c, gathered)
        d_s = fr.DeviceBuffer(c, max(hi - lo, 1) * 4)
        wk.fc_from_slices_lp(B, lo, hi - lo, d_g, fr.FC_FP32, d_s)
        wk.sync()
        scores[lo:hi] = d_s.download(np.float32, hi - lo)
        with pytest.raises(fr.FleetRecError):
            wk.fc_from_slices_lp(B, lo, hi - lo, d_g, fr.FC_BF16, d_s)   # low-precision transports are device paths
    assert np.array_equal(scores, ref)
    for c, wk in shards:
        wk.close()
        c.close()
    w0.close()
    whole.close()


def test_driver_loops_on_the_cpu(fr, O):
    """main() + the thread_consume() batch loop (cuda_server.c:23-25,406-497,554-560) over a CPU context: THREAD_NUM threads drawing batch
    ids from the mutex-guarded counter, resident and host-buffer forms; the scores in the rings are the worker's own."""
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=5000)
    ctx = fr.Context(m, device=CPU)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    rng = np.random.default_rng(3)
    B = 32
    host = [uniform_idx(rng, m.rows(), B) for _ in range(4)]
    pool = [fr.DeviceBuffer.from_numpy(ctx, a) for a in host]
    wk = fr.Worker(ctx, B)
    want = [wk.infer(a) for a in host]
    dv = fr.Driver(ctx, 2, 2, B)
    assert dv.run_resident(B, 24, pool) > 0
    ring = dv.score_ring(0, 0, B)
    assert any(any(np.array_equal(ring[k], w_) for w_ in want) for k in range(4))
    assert dv.run_host(B, 24, host) > 0
    with pytest.raises(fr.FleetRecError):
        dv.run_host(B, 8, host, streaming=True)
    dv.close()
    wk.close()
    ctx.close()


def test_server_answers_the_sender_on_the_cpu_back_end(fr):
    """BASELINE configs[0] end to end -- "Model-A batch = 1 ... on host CPU (plumbing, no accelerator)": fleetrec_server --device -1 takes
    batches of ONE item from fleetrec_sender over TCP, with the reference's data (even/odd tables, the 32 fixed indices, all-ones weights):
    every thread's last score is K * H1 * H2 * H3 or 0 (cuda_server.c:499-502 prints the first five of the last batch; a batch of one has one)."""
    if not os.path.exists(os.path.join(HOST, "fleetrec_server")):
        subprocess.check_call(["make", "-s", "-C", HOST])
    for batch, total in ((1, 64), (128, 16)):
        threads = 2
        port = free_port_block(threads)
        srv = subprocess.Popen([os.path.join(HOST, "fleetrec_server"), "--model", "A", "--batch", str(batch), "--threads", str(threads), "--port", str(port),
                                "--total", str(total), "--tables", "evenodd", "--weights", "ones", "--device", "-1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        time.sleep(0.5)
        snd = subprocess.Popen([os.path.join(HOST, "fleetrec_sender"), "--model", "A", "--batch", str(batch), "--threads", str(threads), "--port", str(port),
                                "--indices", "reference"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        try:
            out, _ = srv.communicate(timeout=120)
            snd.communicate(timeout=60)
        finally:
            for p in (srv, snd):
                if p.poll() is None:
                    p.kill()
        out = out.decode()
        assert srv.returncode == 0, out
        assert "processed %d batches" % total in out, out
        rows = re.findall(r"thread \d+ scores:((?: [-0-9.e+]+)+)", out)
        assert rows, out
        val = 352.0 * 2 ** 27
        for r in rows:
            v = [float(x) for x in r.split()]
            assert v == ([0.0, 0.0, val, val, 0.0] if batch >= 5 else v) and all(x in (0.0, val) for x in v), (v, out)


@pytest.mark.parametrize("seed", range(6))
def test_random_custom_models(fr, seed):
    """User-defined models (the run-time counterpart of the reference's generated constants.hpp) on the CPU back-end: random tables,
    an optional dense block, COPY pads, random FC widths; tables and weights uploaded from the host.  The record against its semantic
    definition (concatenate the segments), bit for bit; the scores against an fp64 chain of the same weights."""
    import gpu_helpers as helpers     # (the random-model generator the GPU suite uses: tests/gpu_helpers.py)
    rng = np.random.default_rng(1000 + seed)
    m, segs, fcw = helpers._random_model(fr, rng, 64 if seed % 2 == 0 else 32)
    ctx = fr.Context(m, device=CPU)
    host = [rng.standard_normal((t.rows, t.dim)).astype(np.float32) for t in m.tables()]
    for t, a in enumerate(host):
        ctx.upload_table(t, a)
        assert np.array_equal(ctx.download_table(t, 0, a.shape[0], dtype=np.float32), a)
    ws = [(rng.uniform(-1, 1, fcw[i] * fcw[i + 1]) / np.sqrt(fcw[i])).astype(np.float32) for i in range(4)]
    for l in range(4):
        ctx.set_weights(l, ws[l])
        assert np.array_equal(ctx.get_weights(l), ws[l])
    B = int(rng.integers(1, 300))
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    want = np.empty((B, m.record_len), np.float32)
    for (k, src, c0, off, ln, _) in segs:
        want[:, off:off + ln] = dense[:, c0:c0 + ln] if k == fr.SEG_DENSE else host[src][idx[:, src], c0:c0 + ln]
    wk = fr.Worker(ctx, B)
    assert np.array_equal(wk.gather_records(idx, dense).reshape(B, m.record_len), want.view(np.uint32))
    x = want.astype(np.float64)
    for l in range(4):   # column-major H x K: element (h, k) at w[h + k * H]  ->  [k][h]
        x = (x @ ws[l].astype(np.float64).reshape(fcw[l], fcw[l + 1])).astype(np.float32).astype(np.float64)
    got = wk.infer(idx, dense)
    assert rel_err(got, x[:, 0].astype(np.float32)) <= 2e-6 or B < 8      # (a handful of items: max|ref| is noise)
    wk.close()
    ctx.close()


def test_a_forked_child_gets_a_thread_pool_of_its_own(fr):
    """fork() (Python's multiprocessing) copies the CPU back-end's pool object but none of its helper threads: a parallel region in the child would
    wait for helpers that do not exist.  The child's atfork handler drops the inherited pool; its first call builds its own."""
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=5000)
    ctx = fr.Context(m, device=CPU)
    ctx.fill_tables(fr.FILL_HASH, 1)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    wk = fr.Worker(ctx, 512)
    idx = uniform_idx(np.random.default_rng(0), m.rows(), 512)
    fr.cpu_set_threads(4)
    ref = wk.infer(idx)                       # the parent's pool is up and has run
    pid = os.fork()
    if pid == 0:
        try:
            ok = np.array_equal(wk.infer(idx), ref)
        except BaseException:                  # noqa: BLE001
            ok = False
        os._exit(0 if ok else 3)
    deadline = time.time() + 60
    while time.time() < deadline:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        pytest.fail("the forked child hung in a parallel region")
    assert os.WEXITSTATUS(status) == 0
    assert np.array_equal(wk.infer(idx), ref)
    fr.cpu_set_threads(0)
    wk.close()
    ctx.close()


# ---- the table-sharded step behind the C-ABI with G > 1 ranks: fr_comm_init_all over G CPU shard contexts = the in-process host exchange
#      (csrc/fr_comm.cpp: the SAME step / status words / reference counts / bounded wait as over RCCL, another transport below them).
#      Counterpart of the 3-node server taking every batch from three senders (3-node cuda_server.c:513-591). ---------------------------------
def _sharded_job(fr, G, max_batch, max_rows=2000):
    m = fr.Model.builtin(fr.MODEL_C).clone(max_rows=max_rows)
    ctxs, wks = [], []
    for r in range(G):
        c = fr.Context(m, device=CPU, shard_rank=r, n_shards=G)
        c.fill_tables(fr.FILL_HASH, SEED_TABLES)
        c.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
        ctxs.append(c)
        wks.append(fr.Worker(c, max_batch))
    whole = fr.Context(m, device=CPU)
    whole.fill_tables(fr.FILL_HASH, SEED_TABLES)
    whole.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    return m, ctxs, wks, fr.Comm.init_all(ctxs), whole, fr.Worker(whole, max_batch)


def _close_job(ctxs, wks, comms, whole, w0):
    for w in wks:
        w.close()
    for cm in comms:
        cm.close()
    for c in ctxs:
        c.close()
    w0.close()
    whole.close()


def _load_request(wks, idx, dense):
    for w in wks:                       # every rank holds the whole request batch (as for fr_worker_submit)
        w.idx[:len(idx)] = idx
        w.dense[:len(idx)] = dense


def _sync_status(fr, w):
    try:
        w.sync()
        return fr.FR_OK, ""
    except fr.FleetRecError as e:
        return e.status, str(e)


@pytest.mark.parametrize("G,batches", [(2, (301, 64, 1)), (3, (301, 2, 300)), (8, (301, 5, 512))])
def test_sharded_step_with_several_ranks_through_the_host_exchange(fr, G, batches):
    """VERDICT r05 item 1.  G = 2, 3 (uneven: B = 301), 8 ranks on row-capped Model-C: after fr_worker_sync EVERY rank's score buffer holds
    all B scores, bit-identical to an unsharded CPU context's -- for batches that do not divide by G, for batches smaller than G (ranks with
    no items of their own still join both collectives), driven by G host threads (the server's shape) and by ONE thread that submits on
    all ranks and then synchronises them (a CPU worker's step runs on its own host stream behind the call)."""
    import threading
    B_max = max(batches)
    m, ctxs, wks, comms, whole, w0 = _sharded_job(fr, G, B_max)
    rng = np.random.default_rng(100 + G)
    try:
        for step, B in enumerate(batches):
            idx = uniform_idx(rng, m.rows(), B)
            dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
            ref = w0.infer(idx, dense)
            _load_request(wks, idx, dense)
            got = [None] * G
            if step % 2 == 0:           # G threads, one per rank: submit + sync each
                def rank(r):
                    wks[r].submit_sharded(comms[r], B)
                    wks[r].sync()
                    got[r] = wks[r].score[:B].copy()
                ts = [threading.Thread(target=rank, args=(r,)) for r in range(G)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join(120)
                assert not any(t.is_alive() for t in ts)
            else:                       # one thread: all submits, then all syncs (in reverse rank order for good measure)
                for r in range(G):
                    wks[r].submit_sharded(comms[r], B)
                for r in reversed(range(G)):
                    wks[r].sync()
                    got[r] = wks[r].score[:B].copy()
            for r in range(G):
                assert got[r] is not None and np.array_equal(got[r], ref), (G, B, r)
    finally:
        _close_job(ctxs, wks, comms, whole, w0)


def test_sharded_step_failure_protocol_across_ranks(fr):
    """The three kinds of failure of a collective step (csrc/fr_comm.cpp), each with more than one rank for the first time:
    (1) an argument error is returned before anything is enqueued and leaves the communicator usable;
    (2) a failed FC chain on rank q (fleetrec_diag.h's injection hook) -> EVERY rank's fr_worker_sync returns FR_ERR_COMM naming q, q's
        items are NaN on every rank, the others' scores are right, and the next step works again;
    (3) a rank that never arrives trips the bounded wait of the ranks that did (fr_comm_set_wait_ms), the communicator is aborted, every
        later call on any rank of the group answers FR_ERR_COMM."""
    import importlib
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    G, B, q = 3, 301, 1
    m, ctxs, wks, comms, whole, w0 = _sharded_job(fr, G, 400)
    rng = np.random.default_rng(9)
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    ref = w0.infer(idx, dense)
    try:
        _load_request(wks, idx, dense)
        # (1)
        with pytest.raises(fr.FleetRecError) as e:
            wks[0].submit_sharded(comms[0], 401)
        assert e.value.status == fr.FR_ERR_INVALID
        with pytest.raises(fr.FleetRecError) as e:
            wks[0].submit_sharded(comms[1], B)                  # another rank's communicator
        assert e.value.status == fr.FR_ERR_INVALID
        # (2)
        wks[q].inject_fc_failure(1)
        for r in range(G):
            wks[r].submit_sharded(comms[r], B)
        lo, hi = dist_mod.item_range(q, G, B)
        for r in range(G):
            st, text = _sync_status(fr, wks[r])
            assert st == fr.FR_ERR_COMM and "shard rank %d reported a failed FC chain" % q in text, (r, st, text)
            sc = wks[r].score[:B]
            assert np.isnan(sc[lo:hi]).all() and np.array_equal(sc[:lo], ref[:lo]) and np.array_equal(sc[hi:], ref[hi:]), r
        for r in range(G):                                      # the communicator survived: the same request again, nobody fails
            wks[r].submit_sharded(comms[r], B)
        for r in range(G):
            wks[r].sync()
            assert np.array_equal(wks[r].score[:B], ref), r
        # (3): rank 2 never submits
        for cm in comms:
            cm.set_wait_ms(300)
        t0 = time.time()
        for r in (0, 1):
            wks[r].submit_sharded(comms[r], B)
        for r in (0, 1):
            st, text = _sync_status(fr, wks[r])
            assert st == fr.FR_ERR_COMM and ("did not complete within 300 ms" in text or "aborted" in text), (r, st, text)
        assert time.time() - t0 < 20
        for r in range(G):                                      # aborted for good, on every rank of the in-process group
            with pytest.raises(fr.FleetRecError) as e:
                wks[r].submit_sharded(comms[r], B)
            assert e.value.status == fr.FR_ERR_COMM
        st, _ = _sync_status(fr, wks[2])                        # nothing in flight there
        assert st == fr.FR_OK
    finally:
        _close_job(ctxs, wks, comms, whole, w0)


def test_sharded_step_outlives_its_communicator_handle_and_its_worker(fr):
    """Reference counts of a communicator: fr_comm_destroy between the submits and the syncs only drops the handles -- the
    step in flight keeps the communicator (and the group all G handles share) alive until its worker has synchronised, and the scores are
    right.  And a worker destroyed with a step still parked in a rendezvous (its peer never came) does not hang: the destroy waits for the
    communicator's bound, aborts the exchange and joins the host stream."""
    G, B = 2, 77
    m, ctxs, wks, comms, whole, w0 = _sharded_job(fr, G, 128)
    rng = np.random.default_rng(4)
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    ref = w0.infer(idx, dense)
    try:
        _load_request(wks, idx, dense)
        for r in range(G):
            wks[r].submit_sharded(comms[r], B)
        for cm in comms:
            cm.close()                                          # destroy in flight
        for r in range(G):
            wks[r].sync()
            assert np.array_equal(wks[r].score[:B], ref), r
        # a second job on the same contexts: rank 1 never arrives, rank 0's worker is destroyed with its step in flight
        comms2 = fr.Comm.init_all(ctxs)
        comms2[0].set_wait_ms(200)
        wks[0].submit_sharded(comms2[0], B)
        t0 = time.time()
        wks[0].close()
        assert time.time() - t0 < 20
        with pytest.raises(fr.FleetRecError) as e:              # the group was aborted by that destroy
            wks[1].submit_sharded(comms2[1], B)
        assert e.value.status == fr.FR_ERR_COMM
        for cm in comms2:
            cm.close()
    finally:
        _close_job(ctxs, wks, comms, whole, w0)


def test_server_shards_model_c_over_cpu_shard_contexts(fr):
    """The end-to-end form of the G > 1 step: `fleetrec_server --shards 3 --device -1` -- the 3-node server's shape
    (three parts per batch, 3-node cuda_server.c:513-591) with three CPU shard contexts exchanging in process -- fed by fleetrec_sender over
    TCP with the reference's data (even/odd tables, the 32 fixed indices, all-ones weights): the first five scores of every thread's last
    batch are 0 0 K*H1*H2*H3 K*H1*H2*H3 0."""
    if not os.path.exists(os.path.join(HOST, "fleetrec_server")):
        subprocess.check_call(["make", "-s", "-C", HOST])
    threads, total, batch = 2, 12, 100            # 100 items over 3 ranks: 34 + 33 + 33
    port = free_port_block(threads)
    srv = subprocess.Popen([os.path.join(HOST, "fleetrec_server"), "--model", "C", "--batch", str(batch), "--threads", str(threads), "--port", str(port),
                            "--total", str(total), "--tables", "evenodd", "--weights", "ones", "--row-cap", "200", "--shards", "3", "--device", "-1"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    time.sleep(0.5)
    snd = subprocess.Popen([os.path.join(HOST, "fleetrec_sender"), "--model", "C", "--batch", str(batch), "--threads", str(threads), "--port", str(port),
                            "--indices", "reference", "--row-cap", "200"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = srv.communicate(timeout=300)
        snd.communicate(timeout=60)
    finally:
        for p in (srv, snd):
            if p.poll() is None:
                p.kill()
    out = out.decode()
    assert srv.returncode == 0, out
    assert "table-sharded over 3 CPU shard contexts" in out and "processed %d batches" % total in out, out
    rows = re.findall(r"thread \d+ scores:((?: [-0-9.e+]+)+)", out)
    assert rows, out
    val = 3968.0 * 2 ** 28
    for r in rows:
        assert [float(x) for x in r.split()] == [0.0, 0.0, val, val, 0.0], (r, out)


def test_a_context_destroyed_before_its_workers_and_communicators_leaves_nothing_dangling(fr):
    """Lifetime (round 6): live workers and communicator handles hold the context; fr_ctx_destroy first is allowed -- the workers keep working
    and the last object to go releases the tables.  (Before: the worker's destructor walked a freed context; a failed GPU test's teardown
    ended in glibc's "corrupted double-linked list".)  Runs under ASan in tools/run_sanitizers.sh."""
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=2000)
    ctx = fr.Context(m, device=CPU)
    ctx.fill_tables(fr.FILL_HASH, 1)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    wks = [fr.Worker(ctx, 32) for _ in range(3)]
    idx = np.zeros((32, m.n_tables), np.int32)
    want = wks[0].infer(idx)
    ctx.close()                                   # the handle goes first
    for w in wks:
        assert np.array_equal(w.infer(idx), want)  # ... and the workers still have their context
        w.close()
    # sharded: contexts destroyed before their communicator handles and workers
    mc = fr.Model.builtin(fr.MODEL_C).clone(max_rows=500)
    ctxs = [fr.Context(mc, device=CPU, shard_rank=r, n_shards=2) for r in range(2)]
    for c in ctxs:
        c.fill_tables(fr.FILL_HASH, 1)
        c.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    w2 = [fr.Worker(c, 16) for c in ctxs]
    comms = fr.Comm.init_all(ctxs)
    for c in ctxs:
        c.close()
    rng = np.random.default_rng(1)
    ix = uniform_idx(rng, mc.rows(), 16)
    de = rng.uniform(-1, 1, (16, mc.dense_len)).astype(np.float32)
    _load_request(w2, ix, de)
    for r in range(2):
        w2[r].submit_sharded(comms[r], 16)
    for r in range(2):
        w2[r].sync()
    assert np.array_equal(w2[0].score[:16], w2[1].score[:16])
    for cm in comms:
        cm.close()
    for w in w2:
        w.close()
