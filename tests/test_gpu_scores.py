"""GPU parity, part 2: fp32 scores against the oracle, the reference's known answers, error reporting, user models, committed fixtures.
Tolerances, seeds and reference chains: tests/gpu_helpers.py; the full-size contexts (`ctxs`): tests/conftest.py."""
import os
import threading

import numpy as np
import pytest
from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", [0, 1, 2])
def test_reference_literal_mode_known_answer(fr, O, gpu, which):
    """The reference's own run: even/odd tables, ONE index per item broadcast to every table
    (FR_INDEX_PER_ITEM), the 32 fixed indices, all-ones weights -> K*H1*H2*H3 or 0, exactly."""
    base = fr.Model.builtin(which)
    m = base.clone(max_rows=200, index_mode=fr.INDEX_PER_ITEM)  # host.cpp initialises 200 rows (DEBUG)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_EVEN_ODD, 0)
    ctx.fill_weights(fr.WEIGHTS_ONES, 0)
    idx = np.tile(om.halves[0].idx_random, 3)  # batch_num = 3
    B = len(idx)
    dense = np.tile(np.where(idx % 2 == 0, 1.0, 0.0).astype(np.float32)[:, None], (1, m.dense_len)) if m.dense_len else None
    wk = fr.Worker(ctx, B)
    rec = wk.gather_records(idx[:, None], dense).reshape(B, m.record_len)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_EVEN_ODD)
    assert np.array_equal(rec, want)
    assert all((rec[j] == (0x3F800000 if idx[j] % 2 == 0 else 0)).all() for j in range(B))
    scores = wk.infer(idx[:, None], dense)
    fc = m.fc
    val = np.float32(float(fc[0]) * fc[1] * fc[2] * fc[3])
    assert np.array_equal(scores, np.where(idx % 2 == 0, val, np.float32(0)))
    wk.close()
    ctx.close()


def test_readme_known_answers(fr, gpu):
    """GPU/final_network_cublasLt_1_node_no_FIFO_scatter/README.md:7-11: K=512 -> 2^36, K=1024 -> 2^37."""
    for K, want in ((512, 2.0 ** 36), (1024, 2.0 ** 37)):
        # a one-table model whose record is K floats: 1 table of dim K? rows must be dim<=1024 multiple of 4
        T = fr.TableDesc(mem_class=0, table_id=0, source=0, dim=K, rows=4, bank=0, round=0, addr_axi=0)
        S = fr.Segment(kind=fr.SEG_TABLE, src=0, src_col=0, rec_offset=0, len=K, source=0)
        d = fr.ModelDesc()
        d.name = b"readme"
        d.n_tables, d.n_segments = 1, 1
        import ctypes
        d.tables = ctypes.pointer(T)
        d.segments = ctypes.pointer(S)
        d.record_len, d.dense_len = K, 0
        for i, v in enumerate((K, 1024, 512, 256, 1)):
            d.fc[i] = v
        m = fr.Model(ctypes.pointer(d), keepalive=(T, S, d))
        ctx = fr.Context(m, device=gpu)
        ctx.upload_table(0, np.ones((4, K), np.float32))
        ctx.fill_weights(fr.WEIGHTS_ONES, 0)
        wk = fr.Worker(ctx, 128)  # BATCH_SIZE 128 (constant.h:32)
        s = wk.infer(np.zeros((128, 1), np.int32))
        assert (s == np.float32(want)).all()
        wk.close()
        # the same closed form is exact in the low-precision chains: every operand and every activation (K, K*2^10, K*2^19) is a
        # power of two, representable in bf16 and -- once the calibration batch has set the exponents -- in e4m3
        for prec in (fr.FC_BF16, fr.FC_FP8):
            ctx.set_fc_precision(prec)
            wk = fr.Worker(ctx, 128)
            if prec == fr.FC_FP8:
                wk.calibrate_fp8(np.zeros((128, 1), np.int32))
            s = wk.infer(np.zeros((128, 1), np.int32))
            assert (s == np.float32(want)).all(), (prec, s[:4])
            wk.close()
        ctx.close()


@pytest.mark.parametrize("which,B", [(0, 256), (1, 1024), (2, 512)])
def test_scores_within_tolerance(fr, O, ctxs, which, B):
    """End-to-end submit()/sync() through pinned host buffers vs the oracle chain (fp64 accumulate)."""
    m, ctx = ctxs(which)
    om = O.OracleModel(NAMES[which])
    rng = np.random.default_rng(77)
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    wk = fr.Worker(ctx, B)
    scores = wk.infer(idx, dense)
    rec = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    ws = [ctx.get_weights(l) for l in range(4)]
    ref = om.fc_chain(rec.view(np.float32), ws, acc64=True)
    e = rel_err(scores, ref)
    assert e <= 1e-3, e
    assert e <= 2e-5, "fp32 MFMA path should be far inside the tolerance (got %g)" % e
    assert rel_err_each(scores, ref) <= 1e-3, rel_err_each(scores, ref)     # north_star's wording, item by item
    # the pipeline's own feature-major gather (gather_t) is bit-exact too: Xt[k][m] == record[m][k]
    assert np.array_equal(wk.features(B), rec.T)
    # fc_only on oracle records gives the same scores as the fused pipeline (bitwise: same FC kernels)
    assert np.array_equal(wk.fc_scores(rec.view(np.float32)), scores)
    # deterministic: the same batch again is bitwise identical (fixed-order split-K sums, no atomics)
    assert np.array_equal(wk.infer(idx, dense), scores)
    # ragged batch sizes (the split-K factor, hence the fp32 summation order, may differ with the batch size)
    for b in (1, 3, 63, 65):
        s_b = wk.infer(idx[:b], None if dense is None else dense[:b])
        assert np.abs(s_b - scores[:b]).max() <= 1e-5 * np.abs(ref).max()
    wk.close()


def test_blocked_layout_matches_3node_buffer(fr, O, gpu):
    """FR_LAYOUT_BLOCKED reproduces the 3-node server's receive buffer [CPU][FPGA0][FPGA1]
    (3-node cuda_server.c:515,541,566) and the FC then reads it as B x 3968 item-major (F8)."""
    m = fr.Model.builtin(fr.MODEL_C).clone(max_rows=3000, layout=fr.LAYOUT_BLOCKED)
    om = O.OracleModel("C")
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, 3)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, 4)
    rng = np.random.default_rng(9)
    B = 96
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, 64)).astype(np.float32)
    wk = fr.Worker(ctx, B)
    got = wk.gather_records(idx, dense)
    sem = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=3)
    blk = om.block_records(sem)
    assert np.array_equal(got, blk)
    scores = wk.infer(idx, dense)
    ws = [ctx.get_weights(l) for l in range(4)]
    ref = om.fc_chain(blk.view(np.float32).reshape(B, m.record_len), ws, acc64=True)
    assert rel_err(scores, ref) <= 1e-3
    wk.close()
    # a ragged batch >= 1024: the blocked destination ([source block][item][word]: dst_blk x batch + dst_off) through gather_pack_stream_kernel
    B2 = 1024 + 29
    idx2 = uniform_idx(rng, m.rows(), B2)
    dense2 = rng.uniform(-1, 1, (B2, 64)).astype(np.float32)
    wk2 = fr.Worker(ctx, B2)
    assert np.array_equal(wk2.gather_records(idx2, dense2), om.block_records(om.gather(idx2, dense=dense2, content_mode=O.FILL_HASH, seed=3)))
    wk2.close()
    ctx.close()


def test_nan_in_a_table_reaches_the_score(fr, gpu):
    """The reference's fp32 chain has no guard: a NaN in a looked-up row makes that item's score a NaN and nobody else's.  Same here in
    all three precisions and on both paths (unpipelined submit, fused streaming kernels) -- the bf16 conversion keeps a NaN a NaN and
    the fp8 saturation is done with compares so that a NaN is not clamped into -448."""
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=500)
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    t_bad, r_bad = 11, 7
    dim = m.tables()[t_bad].dim
    row = ctx.download_table(t_bad, r_bad, 1, dtype=np.float32).reshape(1, dim).copy()
    row[0, 1] = np.nan
    ctx.upload_table(t_bad, row, row0=r_bad)
    rng = np.random.default_rng(5)
    B = 256
    idx = uniform_idx(rng, m.rows(), B)
    idx[idx[:, t_bad] == r_bad, t_bad] = r_bad + 1      # nobody hits the bad row ...
    clean = idx.copy()
    hit = [3, 64, 200]
    idx[hit, t_bad] = r_bad                             # ... except these items
    wk = fr.Worker(ctx, B)
    rec = wk.gather_records(idx).reshape(B, m.record_len).view(np.float32)
    assert np.isnan(rec[hit]).sum() == len(hit) and not np.isnan(np.delete(rec, hit, axis=0)).any()
    d_idx = fr.DeviceBuffer.from_numpy(ctx, idx)
    d_sc = fr.DeviceBuffer(ctx, B * 4)
    for prec in (fr.FC_FP32, fr.FC_BF16, fr.FC_FP8):
        ctx.set_fc_precision(prec)
        if prec == fr.FC_FP8:
            wk.calibrate_fp8(clean)
        for path in ("submit", "push"):
            if path == "submit":
                s = wk.infer(idx)
            else:
                wk.push_device(B, d_idx, None, d_sc)
                wk.sync()
                s = d_sc.download(np.float32, B)
            assert np.isnan(s[hit]).all(), (prec, path, s[hit])
            assert np.isfinite(np.delete(s, hit)).all(), (prec, path)
    wk.close()
    ctx.close()


def test_index_out_of_range_is_reported(fr, gpu):
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=1000)
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, 1)
    ctx.fill_weights(fr.WEIGHTS_ONES, 0)
    wk = fr.Worker(ctx, 8)
    idx = np.zeros((8, m.n_tables), np.int32)
    wk.infer(idx)  # fine
    idx[5, 17] = m.rows()[17]  # == rows -> out of range
    with pytest.raises(fr.FleetRecError) as e:
        wk.infer(idx)
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    idx[5, 17] = -1
    with pytest.raises(fr.FleetRecError):
        wk.infer(idx)
    idx[5, 17] = 0
    wk.infer(idx)  # the flag is cleared after it has been reported
    # API misuse is rejected loudly
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit(9)
    assert e.value.status == fr.FR_ERR_INVALID
    wk.submit(8)
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit(8)  # one batch in flight per worker
    assert e.value.status == fr.FR_ERR_STATE
    wk.sync()
    wk.close()
    ctx.close()


def test_state_errors(fr, gpu):
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=1000)
    ctx = fr.Context(m, device=gpu)
    wk = fr.Worker(ctx, 8)
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit(8)  # tables not filled
    assert e.value.status == fr.FR_ERR_STATE
    ctx.fill_tables(fr.FILL_EVEN_ODD, 0)
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit(8)  # weights not set
    assert e.value.status == fr.FR_ERR_STATE
    with pytest.raises(fr.FleetRecError):
        ctx.set_weights(0, np.ones(7, np.float32))
    with pytest.raises(fr.FleetRecError):
        fr.Context(m, device=99)
    for bad in (0, -3, 1 << 25):
        with pytest.raises(fr.FleetRecError) as e:
            fr.Worker(ctx, bad)   # empty / negative / absurd batch capacity
        assert e.value.status == fr.FR_ERR_INVALID
    with pytest.raises(fr.FleetRecError) as e:
        fr.Worker(ctx, 1 << 21)   # 1024 x 2 Mi floats: an activation tensor would not fit 32-bit buffer offsets
    assert e.value.status == fr.FR_ERR_INVALID and "4 GiB" in str(e.value)
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit(0)              # empty batch
    assert e.value.status == fr.FR_ERR_INVALID
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit(9)              # more than the worker's capacity
    assert e.value.status == fr.FR_ERR_INVALID
    with pytest.raises(fr.FleetRecError) as e:
        ctx.set_fc_precision(7)
    assert e.value.status == fr.FR_ERR_INVALID
    wk.close()
    ctx.close()


def test_upload_set_weights_roundtrip(fr, O, gpu):
    """User-supplied tables and weights (host.cpp:739-749 migrate; cuda_server.c:346-354 weights H2D)."""
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=257)
    om = O.OracleModel("A")
    ctx = fr.Context(m, device=gpu)
    rng = np.random.default_rng(11)
    host_tabs = []
    for t, d in enumerate(m.tables()):
        a = rng.standard_normal((d.rows, d.dim)).astype(np.float32)
        ctx.upload_table(t, a)
        host_tabs.append(a)
    ws = [(rng.uniform(-1, 1, m.fc[i] * m.fc[i + 1]) / np.sqrt(m.fc[i])).astype(np.float32) for i in range(4)]
    for l in range(4):
        ctx.set_weights(l, ws[l])
        assert np.array_equal(ctx.get_weights(l), ws[l])
    assert np.array_equal(ctx.download_table(3, 5, 90, dtype=np.float32), host_tabs[3][5:95])
    B = 100
    idx = uniform_idx(rng, m.rows(), B)
    wk = fr.Worker(ctx, B)
    rec = wk.gather_records(idx).reshape(B, -1).view(np.float32)
    want = np.concatenate([host_tabs[t][idx[:, t]] for t in range(m.n_tables)], axis=1)  # tables are in wire order
    assert np.array_equal(rec.view(np.uint32), want.view(np.uint32))
    scores = wk.infer(idx)
    ref = om.fc_chain(want, ws, acc64=True)
    assert rel_err(scores, ref) <= 1e-3
    wk.close()
    ctx.close()


def test_concurrent_workers(fr, O, ctxs):
    """THREAD_NUM host threads, each with its own worker/stream, sharing one context
    (cuda_server.c:554-556): results identical to the single-worker run."""
    m, ctx = ctxs(fr.MODEL_A)
    rng = np.random.default_rng(3)
    B, n_threads, n_batches = 256, 4, 6
    idx = [uniform_idx(rng, m.rows(), B) for _ in range(n_threads * n_batches)]
    wk0 = fr.Worker(ctx, B)
    expect = [wk0.infer(i) for i in idx]
    wk0.close()
    results = [None] * len(idx)
    errors = []

    def run(tid):
        try:
            wk = fr.Worker(ctx, B)
            for j in range(n_batches):
                k = tid * n_batches + j
                results[k] = wk.infer(idx[k])
            wk.close()
        except Exception as ex:  # pragma: no cover
            errors.append(ex)

    th = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors
    for r, e in zip(results, expect):
        assert np.array_equal(r, e)


def test_size_independent_properties(fr, ctxs):
    """Full-size Model-C, batch 4096 (BASELINE config 4 shape): properties that need no oracle --
    permutation equivariance, idempotence, and linearity of the score in the dense features."""
    m, ctx = ctxs(fr.MODEL_C)
    rng = np.random.default_rng(2024)
    B = 4096
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    wk = fr.Worker(ctx, B)
    rec = wk.gather_records(idx, dense).reshape(B, m.record_len)
    perm = rng.permutation(B)
    rec_p = wk.gather_records(idx[perm], dense[perm]).reshape(B, m.record_len)
    assert np.array_equal(rec_p, rec[perm])                               # item order is carried through
    assert np.array_equal(wk.gather_records(idx, dense).reshape(B, -1), rec)  # idempotent
    # checksum of checksums: per-table column sums of the record == sums of the fetched rows (table 0 and the last)
    t0 = m.tables()[0]
    seg = [s for s in m.segments() if s.kind == fr.SEG_TABLE and s.src == 0][0]
    rows = np.concatenate([ctx.download_table(0, int(i), 1) for i in idx[:64, 0]])
    assert np.array_equal(rec[:64, seg.rec_offset:seg.rec_offset + t0.dim], rows)
    s1 = wk.infer(idx, dense)
    s0 = wk.infer(idx, np.zeros_like(dense))
    s2 = wk.infer(idx, 2 * dense)
    # the chain is linear (no activation on the reference path): s(2d) - s(0) == 2 (s(d) - s(0))
    scale = np.abs(s1).max()
    assert np.abs((s2 - s0) - 2 * (s1 - s0)).max() <= 1e-4 * scale
    assert np.array_equal(wk.infer(idx[perm], dense[perm]), s1[perm])
    wk.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_custom_models(fr, O, gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    m, segs, fcw = _random_model(fr, rng, 64 if seed % 2 == 0 else 32)   # even seeds: hidden widths the fp8 chain accepts
    ctx = fr.Context(m, device=gpu)
    tabs = m.tables()
    host = [rng.standard_normal((t.rows, t.dim)).astype(np.float32) for t in tabs]
    for t, a in enumerate(host):
        ctx.upload_table(t, a)
    ws = [(rng.uniform(-1, 1, fcw[i] * fcw[i + 1]) / np.sqrt(fcw[i])).astype(np.float32) for i in range(4)]
    for l in range(4):
        ctx.set_weights(l, ws[l])
    B = int(rng.integers(1, 300))
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    # semantic definition of the record: concatenate the segments
    want = np.empty((B, m.record_len), np.float32)
    for (k, src, c0, off, ln, _) in segs:
        if k == fr.SEG_DENSE:
            want[:, off:off + ln] = dense[:, c0:c0 + ln]
        else:
            want[:, off:off + ln] = host[src][idx[:, src], c0:c0 + ln]
    wk = fr.Worker(ctx, B)
    got = wk.gather_records(idx, dense).reshape(B, m.record_len)
    assert np.array_equal(got, want.view(np.uint32))
    # the XCD partition of the word-major gather (records of >= 64 words have one): covers the record, cuts between source rows only;
    # and a batch large enough to take that kernel (>= 1024 items of a >= 512-word record) still gathers the same records
    n_words = m.record_len // 4
    if n_words >= 64:
        st = ctx.gather_groups()
        assert st[0] == 0 and st[8] == n_words and all(0 <= st[g + 1] - st[g] <= 256 for g in range(8)), st
        cuts = {0, n_words}
        for (k, src, c0, off, ln, _) in segs:
            cuts.add(off // 4)
            if k == fr.SEG_DENSE:
                cuts.update(range(off // 4, (off + ln) // 4 + 1, 8))
        assert set(st) <= cuts, (st, sorted(set(st) - cuts))
    else:
        with pytest.raises(fr.FleetRecError):
            ctx.gather_groups()
    if n_words >= 512:
        B2 = 1024 + 77
        idx2 = uniform_idx(rng, m.rows(), B2)
        dense2 = rng.uniform(-1, 1, (B2, m.dense_len)).astype(np.float32) if m.dense_len else None
        want2 = np.empty((B2, m.record_len), np.float32)
        for (k, src, c0, off, ln, _) in segs:
            want2[:, off:off + ln] = dense2[:, c0:c0 + ln] if k == fr.SEG_DENSE else host[src][idx2[:, src], c0:c0 + ln]
        wk2 = fr.Worker(ctx, B2)
        assert np.array_equal(wk2.gather_records(idx2, dense2).reshape(B2, m.record_len), want2.view(np.uint32))
        wk2.close()
    scores = wk.infer(idx, dense)
    assert np.array_equal(wk.features(B), want.view(np.uint32).T)
    ref = O.OracleModel("A").fc_chain(want, ws, acc64=True, dims=fcw)
    assert rel_err(scores, ref) <= 1e-3
    # the same model in the bf16 chain when its widths allow it (multiples of 16)
    if all(v % 16 == 0 for v in fcw[:4]):
        ctx.set_fc_precision(fr.FC_BF16)
        w2 = fr.Worker(ctx, B)
        assert rel_err(w2.infer(idx, dense), chain_bf16_reference(want, ws, fcw)) <= 5e-3
        w2.close()
    # ... and in the fp8 chain (hidden widths multiples of 64; the record is zero-padded to 64 k inside the q16 image)
    if all(v % 64 == 0 for v in fcw[1:4]):
        ctx.set_fc_precision(fr.FC_FP8)
        w3 = fr.Worker(ctx, B)
        w3.calibrate_fp8(idx, dense)
        act_exp, w_exp = ctx.fp8_exponents()
        s8 = w3.infer(idx, dense)
        feat = w3.features(B, fp8=True)
        assert np.array_equal(feat[:m.record_len], e4m3_encode(want * np.float32(2.0 ** act_exp[0])).T) and not feat[m.record_len:].any()
        assert rel_err(s8, chain_fp8_reference(want, ws, fcw, act_exp, w_exp)) <= 3e-2
        w3.close()
    else:
        with pytest.raises(fr.FleetRecError):
            ctx.set_fc_precision(fr.FC_FP8)
    wk.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2])
def test_committed_fc_fixtures_on_device(fr, O, ctxs, which):
    """tests/golden/fc_cases_*.npz (made by tests/golden/make_fc_cases.py): the device's procedural weights are the
    fixture's weights bit for bit, its gathered records are the fixture's records, and the scores of the committed index rows land on
    the COMMITTED float64 numbers -- through fr_worker_submit, the streaming push (fused item-tile kernels for A / B, stage pipeline +
    GEMM kernels for C) and fr_worker_fc_only, in all three precisions: 1e-3 f32 (BASELINE.json's tolerance; measured ~1e-6), 3e-2 bf16,
    0.15 fp8, relative to max|expected|."""
    import hashlib
    m, ctx = ctxs(which)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fc_cases_%s.npz" % NAMES[which]))
    assert SEED_TABLES == int(fx["seed_tables"]) and SEED_WEIGHTS == int(fx["seed_weights"])
    idx, dense, ref = fx["idx"], (fx["dense"] if m.dense_len else None), fx["expected"]
    n = len(idx)
    for l in range(4):
        assert hashlib.sha256(ctx.get_weights(l).tobytes()).hexdigest() == str(fx["weights_sha256"][l]), l
    scale = np.abs(ref).max()
    wk = fr.Worker(ctx, n)
    rec = wk.gather_records(idx, dense).reshape(n, -1)
    assert hashlib.sha256(rec.tobytes()).hexdigest() == str(fx["records_sha256"])
    wk.close()
    d_i = fr.DeviceBuffer.from_numpy(ctx, idx)
    d_d = fr.DeviceBuffer.from_numpy(ctx, dense) if dense is not None else None
    try:
        for prec, enum, tol in (("f32", fr.FC_FP32, 1e-3), ("bf16", fr.FC_BF16, 3e-2), ("fp8", fr.FC_FP8, 0.15)):
            ctx.set_fc_precision(enum)
            wk = fr.Worker(ctx, n)
            if prec == "fp8":
                wk.calibrate_fp8(idx, dense)
            got = wk.infer(idx, dense)
            assert np.abs(got - ref).max() <= tol * scale, (prec, "submit", np.abs(got - ref).max() / scale)
            if prec == "f32":
                assert np.abs(got - ref).max() <= 2e-5 * scale      # what the exact-f32 MFMA chain actually reaches
                assert rel_err_each(got, ref) <= 1e-3, rel_err_each(got, ref)      # definition (2), against the committed float64 numbers
            got = wk.fc_scores(rec.view(np.float32))
            assert np.abs(got - ref).max() <= tol * scale, (prec, "fc_only", np.abs(got - ref).max() / scale)
            if prec == "f32":
                assert rel_err_each(got, ref) <= 1e-3
            outs = [fr.DeviceBuffer(ctx, n * 4) for _ in range(3)]
            for o in outs:
                wk.push_device(n, d_i, d_d, o)
            wk.sync()
            for o in outs:
                got = o.download(np.float32, n)
                assert np.abs(got - ref).max() <= tol * scale, (prec, "push", np.abs(got - ref).max() / scale)
                if prec == "f32":
                    assert rel_err_each(got, ref) <= 1e-3
                o.free()
            wk.close()
    finally:
        ctx.set_fc_precision(fr.FC_FP32)
        d_i.free()
        if d_d is not None:
            d_d.free()


@pytest.mark.gpu
def test_environment_cannot_change_a_score(fr, ctxs):
    """the shipped library reads no environment variable -- the experiment knobs (among them FR_GEMM_ABLATE, whose
    values make fc_gemm_pipe_kernel compute WRONG results on purpose) are compiled into libfleetrec_exp.so only.  Model-C batch 4096 bf16
    (the path through that GEMM kernel) and Model-B bf16 through the fused kernels, with the variables set to their most destructive
    values: bit-identical scores."""
    if os.path.basename(fr.LIB_PATH) != "libfleetrec.so":
        pytest.skip("FR_LIB points at another build")
    rng = np.random.default_rng(5)
    knobs = {"FR_GEMM_ABLATE": "2", "FR_GEMM_ORDER": "0", "FR_GEMM_PRIO": "0", "FR_GEMM_PIPE": "0", "FR_LP_GEMM": "0", "FR_GATHER_STREAM": "0", "FR_GATHER_ITEMS": "16",
             "FR_GATHER_XCD": "0", "FR_FUSED": "0", "FR_FUSED_HK": "0", "FR_FUSED_GROUP": "1", "FR_SUBMIT_ZEROCOPY": "0", "FR_GATHER_TR": "0"}
    for which, B in ((2, 4096), (1, 1024)):
        m, ctx = ctxs(which)
        idx = uniform_idx(rng, m.rows(), B)
        dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
        ctx.set_fc_precision(fr.FC_BF16)
        try:
            def run():
                wk = fr.Worker(ctx, B)
                a = wk.infer(idx, dense)
                d_i = fr.DeviceBuffer.from_numpy(ctx, idx)
                d_d = fr.DeviceBuffer.from_numpy(ctx, dense) if dense is not None else None
                o = fr.DeviceBuffer(ctx, B * 4)
                wk.push_device(B, d_i, d_d, o)
                wk.sync()
                b = o.download(np.float32, B)
                for x in (d_i, d_d, o):
                    if x is not None:
                        x.free()
                wk.close()
                return a, b
            base = run()
            old = {k: os.environ.get(k) for k in knobs}
            os.environ.update(knobs)
            try:
                again = run()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            assert np.array_equal(base[0], again[0]) and np.array_equal(base[1], again[1]), which
        finally:
            ctx.set_fc_precision(fr.FC_FP32)
