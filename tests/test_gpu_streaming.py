"""GPU parity, part 3: the streaming forms of the hot loop (push_device, host-fed blocks, the native drivers) against submit / sync.
Tolerances, seeds and reference chains: tests/gpu_helpers.py; the full-size contexts (`ctxs`): tests/conftest.py."""
import os
import threading

import numpy as np
import pytest
from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_streaming_push_matches_submit(fr, O, ctxs):
    """fr_worker_push_device: pushed batches are queued and launched a group at a time through the fused item-tile kernels (32-item
    kernel for small launches, 64-item kernel for launches that cover the chip).  Scores must equal the unpipelined submit()/sync()
    results to 1e-5, and bit for bit from run to run, whatever the batch sizes, the sync points, the group size and the kernel."""
    m, ctx = ctxs(fr.MODEL_A)
    rng = np.random.default_rng(31)
    sizes = [256, 256, 1, 37, 256, 64, 200, 256, 255, 256, 33, 256]
    idx = [uniform_idx(rng, m.rows(), b) for b in sizes]
    wk = fr.Worker(ctx, 256)
    expect = [wk.infer(i) for i in idx]
    d_idx = [fr.DeviceBuffer.from_numpy(ctx, i) for i in idx]
    d_sc = [fr.DeviceBuffer(ctx, 256 * 4) for _ in sizes]
    for cut in (len(sizes), 1, 3, 7):  # sync after `cut` pushes, then push the rest
        for b in d_sc:
            b.upload(np.full(256, np.nan, np.float32))
        for j, b in enumerate(sizes):
            wk.push_device(b, d_idx[j], None, d_sc[j])
            if j + 1 == cut:
                wk.sync()
        wk.sync()
        for j, b in enumerate(sizes):
            got = d_sc[j].download(np.float32, 256)
            # Model-A streams through the fused item-tile kernel (full-K sums, no split-K): same arithmetic, another
            # fp32 summation order than the unpipelined stage launches -> equal to ~1e-6, and bitwise run-to-run
            assert np.abs(got[:b] - expect[j]).max() <= 1e-5 * np.abs(expect[j]).max(), (cut, j)
            assert np.isnan(got[b:]).all()
            if cut == len(sizes):
                first_run = first_run if "first_run" in dir() else {}
                first_run[j] = got[:b].copy()
            else:
                assert np.array_equal(got[:b], first_run[j]), (cut, j)
    # the launch-group knob: any group size from 12 up gives the same scores bit for bit (items are independent); smaller groups ride
    # the stage pipeline (one launch per push) and give the unpipelined submit's scores bit for bit (same stage kernels)
    g0 = ctx.stream_group()
    assert g0 >= 1
    for grp in (1, 5, 12, 32):
        ctx.set_stream_group(grp)
        assert ctx.stream_group() in (grp, 1)
        for j, b in enumerate(sizes):
            wk.push_device(b, d_idx[j], None, d_sc[j])
        wk.sync()
        for j, b in enumerate(sizes):
            want = first_run[j] if (grp >= 12 and g0 > 1) else expect[j][:b]
            assert np.array_equal(d_sc[j].download(np.float32, 256)[:b], want), (grp, j)
    # a group change in mid-stream drains the other path first: every batch still comes out as one of the two
    for j, b in enumerate(sizes):
        ctx.set_stream_group(3 if j % 5 < 2 else 64)
        wk.push_device(b, d_idx[j], None, d_sc[j])
    wk.sync()
    for j, b in enumerate(sizes):
        got = d_sc[j].download(np.float32, 256)[:b]
        assert np.array_equal(got, expect[j][:b] if j % 5 < 2 else first_run[j]), j
    ctx.set_stream_group(g0)
    with pytest.raises(fr.FleetRecError):
        ctx.set_stream_group(0)
    # the 64-item kernel (fr_fused_tile_m2_kernel) takes a launch only when its workgroups would cover more than half of the CUs
    # (> 128 tiles of 64 items): 40 queued batches of 256 = 160 tiles.  Same bits as the 32-item kernel that ran everything above.
    if g0 >= 64:
        full = [j for j, b in enumerate(sizes) if b == 256]
        many = [fr.DeviceBuffer(ctx, 256 * 4) for _ in range(40)]
        for k_, buf in enumerate(many):
            wk.push_device(256, d_idx[full[k_ % len(full)]], None, buf)
        wk.sync()
        for k_, buf in enumerate(many):
            assert np.array_equal(buf.download(np.float32, 256), first_run[full[k_ % len(full)]]), k_
        for buf in many:
            buf.free()
    # mixing: a plain submit is refused while pushes are in flight, and works again after sync
    wk.push_device(256, d_idx[0], None, d_sc[0])
    with pytest.raises(fr.FleetRecError) as e:
        wk.submit_device(256, d_idx[1], None, d_sc[1])
    assert e.value.status == fr.FR_ERR_STATE
    wk.sync()
    assert np.array_equal(wk.infer(idx[2]), expect[2])
    # oracle check of one streamed batch (not only self-consistency)
    om = O.OracleModel("A")
    rec = om.gather(idx[4], content_mode=O.FILL_HASH, seed=SEED_TABLES)
    ref = om.fc_chain(rec.view(np.float32), [ctx.get_weights(l) for l in range(4)], acc64=True)
    assert rel_err(d_sc[4].download(np.float32, 256), ref) <= 1e-3
    wk.close()


def test_driver_loop(fr, ctxs):
    """The native THREAD_NUM-thread batch loop (fr_driver_run_resident) completes and leaves correct scores."""
    m, ctx = ctxs(fr.MODEL_A)
    rng = np.random.default_rng(5)
    pool = [fr.DeviceBuffer.from_numpy(ctx, uniform_idx(rng, m.rows(), 256)) for _ in range(4)]
    drv = fr.Driver(ctx, 4, 2, 256)
    el = drv.run_resident(256, 403, pool)
    assert el > 0
    # every worker's score ring holds results of pool batches: each ring slot must equal the scores of ONE of the 4 pool entries
    wk0 = fr.Worker(ctx, 256)
    expect = [wk0.infer(p_.download(np.int32, 256 * m.n_tables).reshape(256, -1)) for p_ in pool]
    wk0.close()
    scale = max(np.abs(e_).max() for e_ in expect)
    checked = 0
    for t in range(4):
        for sl in range(2):
            ring = drv.score_ring(t, sl, 256)
            for row in ring:
                if not row.any():          # slot never used: threads draw batch ids from a shared counter, shares are uneven
                    continue
                assert min(np.abs(row - e_).max() for e_ in expect) <= 1e-5 * scale
                checked += 1
    assert checked == 403                  # every batch's scores are intact in some worker's ring (403 pushes < one trip round the rings)
    el = drv.run_resident(256, 0, pool)  # empty run is fine
    # a long run wraps every ring several times (staggered per-worker syncs): whatever is left in the rings is still a valid result
    drv.run_resident(256, 6000, pool)
    full = 0
    for t in range(4):
        for sl in range(2):
            for row in drv.score_ring(t, sl, 256):
                if row.any():
                    assert min(np.abs(row - e_).max() for e_ in expect) <= 1e-5 * scale
                    full += 1
    assert full >= 8 * 128
    drv.close()


def test_two_contexts_driven_concurrently(fr, gpu):
    """Two contexts of one process -- Model-A in fp32 with launch group 64 and Model-B in bf16 with launch group 8 -- each driven by its
    own native driver loop AT THE SAME TIME (the library keeps the launch group, the LDS attribute and the error state per context /
    per device / per thread, not per process): both loops finish and every score left in their rings is a correct one."""
    import threading
    ma = fr.Model.builtin(fr.MODEL_A).clone(max_rows=50000)
    mb = fr.Model.builtin(fr.MODEL_B).clone(max_rows=50000)
    ca, cb = fr.Context(ma, device=gpu), fr.Context(mb, device=gpu)
    for c in (ca, cb):
        c.fill_tables(fr.FILL_HASH, SEED_TABLES)
        c.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    cb.set_fc_precision(fr.FC_BF16)
    cb.set_stream_group(8)
    assert ca.stream_group() == 64 and cb.stream_group() == 8
    rng = np.random.default_rng(8)
    ia = [uniform_idx(rng, ma.rows(), 256) for _ in range(4)]
    ib = [uniform_idx(rng, mb.rows(), 512) for _ in range(4)]
    wa, wb = fr.Worker(ca, 256), fr.Worker(cb, 512)
    d_ia = [fr.DeviceBuffer.from_numpy(ca, x) for x in ia]
    d_ib = [fr.DeviceBuffer.from_numpy(cb, x) for x in ib]
    sa = [fr.DeviceBuffer(ca, 256 * 4) for _ in ia]
    sb = [fr.DeviceBuffer(cb, 512 * 4) for _ in ib]
    for j in range(4):            # reference results through the same streaming kernels, one context at a time
        wa.push_device(256, d_ia[j], None, sa[j])
        wb.push_device(512, d_ib[j], None, sb[j])
    wa.sync()
    wb.sync()
    ea = [b_.download(np.float32, 256) for b_ in sa]
    eb = [b_.download(np.float32, 512) for b_ in sb]
    wa.close()
    wb.close()
    da, db = fr.Driver(ca, 2, 2, 256), fr.Driver(cb, 2, 2, 512)
    res = {}
    ta = threading.Thread(target=lambda: res.__setitem__("a", da.run_resident(256, 3000, d_ia)))
    tb = threading.Thread(target=lambda: res.__setitem__("b", db.run_resident(512, 600, d_ib)))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert res["a"] > 0 and res["b"] > 0
    assert ca.stream_group() == 64 and cb.stream_group() == 8
    for drv, exp, B in ((da, ea, 256), (db, eb, 512)):
        seen = 0
        for t in range(2):
            for sl in range(2):
                for row in drv.score_ring(t, sl, B):
                    if row.any():
                        assert any(np.array_equal(row, e_) for e_ in exp)     # bit for bit: same kernels, items independent of the launch mix
                        seen += 1
        assert seen >= 100
    da.close(); db.close(); ca.close(); cb.close()


def test_streaming_with_dense_features_on_a_user_model(fr, gpu):
    """A user-defined model WITH dense request features that streams through the fused item-tile kernel (K = 160, hidden widths
    256 / 512 / 256): the dense block travels through every streaming entry point -- device pushes, copying host pushes and the
    zero-copy staging slot (whose dense pointer none of the three reference models exercises: A and B have no dense features, C does
    not stream through the fused kernel) -- and the scores equal the unpipelined submit."""
    spec = {"name": "dense_user", "dense_len": 32, "dense_at": 3, "fc": [256, 512, 256],
            "tables": [{"dim": 8, "rows": 900}, {"dim": 16, "rows": 70}, {"dim": 4, "rows": 5000, "class": "PLRAM"}, {"dim": 32, "rows": 333},
                       {"dim": 64, "rows": 1200, "class": "DDR"}, {"dim": 4, "rows": 17}]}
    m = fr.Model.from_spec(spec)
    assert m.record_len == 160 and m.dense_len == 32
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, 11)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, 12)
    assert ctx.stream_group() > 1          # eligible for the fused streaming path
    rng = np.random.default_rng(13)
    B = 200
    wk = fr.Worker(ctx, B)
    pool = [(uniform_idx(rng, m.rows(), B), rng.uniform(-1, 1, (B, 32)).astype(np.float32)) for _ in range(3)]
    expect = [wk.infer(i_, d_).copy() for i_, d_ in pool]
    scale = max(np.abs(e_).max() for e_ in expect)
    outs = []
    for rep in range(30):
        j, b = rep % 3, [200, 1, 77][rep % 3]
        idx, dense = pool[j]
        out = np.full(B, np.nan, np.float32)
        if rep % 3 == 0:
            d_i, d_d, d_s = fr.DeviceBuffer.from_numpy(ctx, idx[:b]), fr.DeviceBuffer.from_numpy(ctx, dense[:b]), fr.DeviceBuffer(ctx, B * 4)
            wk.push_device(b, d_i, d_d, d_s)
            outs.append(("dev", d_s, j, b, (d_i, d_d)))
        elif rep % 3 == 1:
            wk.push_host(idx[:b], dense[:b], out)
            outs.append(("host", out, j, b, None))
        else:
            si, sd = wk.stage_acquire(b)
            assert sd is not None and sd.shape == (b, 32)
            si[:], sd[:] = idx[:b], dense[:b]
            wk.push_staged(b, out)
            outs.append(("staged", out, j, b, None))
    wk.sync()
    ref_first = {}
    for kind, o, j, b, _ in outs:
        got = o.download(np.float32, B)[:b] if kind == "dev" else o[:b]
        assert np.abs(got - expect[j][:b]).max() <= 1e-5 * scale, (kind, j, b)      # fused kernel: whole-K sums vs the stage launches' split-K order
        key = (j, b)
        if key in ref_first:
            assert np.array_equal(got, ref_first[key]), (kind, j, b)               # and bitwise the same through every entry point
        else:
            ref_first[key] = got.copy()
    with pytest.raises(fr.FleetRecError):
        wk.push_host(pool[0][0], None, np.zeros(B, np.float32))                    # dense features are mandatory for this model
    wk.close()
    ctx.close()


@pytest.mark.parametrize("prec", ["f32", "bf16", "fp8"])
def test_dense_block_through_the_fused_tile_kernels(fr, gpu, prec):
    """ft_gather_tile (the software-pipelined gather phase of the straight-line fused kernels) with a DENSE block in the record: a user
    model of Model-A's shape (K = 352 floats, 1024 / 512 / 256) whose record carries 32 request features between its tables, so that
    it streams through fr_fused_tile_kernel<2, 44, ...> (partial launches), fr_fused_tile_m2_kernel<44> (full launch groups) and the
    bf16 / fp8 64-item kernels -- none of the reference models sends dense words down that path (A and B have none, C does not stream
    through the fused kernels).  Ragged batches; scores against the unpipelined submit of the same rows."""
    rng = np.random.default_rng(31)
    dims = [8, 16, 4, 32, 64, 4, 12, 20, 8, 16, 24, 32, 48, 32]
    assert sum(dims) == 320
    spec = {"name": "dense_352", "dense_len": 32, "dense_at": 5, "fc": [1024, 512, 256],
            "tables": [{"dim": d_, "rows": int(rng.integers(40, 30000))} for d_ in dims]}
    m = fr.Model.from_spec(spec)
    assert m.record_len == 352 and m.dense_len == 32
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, 5)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, 6)
    ctx.set_fc_precision({"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[prec])
    B = 200
    wk = fr.Worker(ctx, B)
    pool = [(uniform_idx(rng, m.rows(), B), rng.uniform(-1, 1, (B, 32)).astype(np.float32)) for _ in range(4)]
    if prec == "fp8":
        wk.calibrate_fp8(pool[0][0], pool[0][1])
    assert ctx.stream_group() == 64
    expect = [wk.infer(i_, d_).copy() for i_, d_ in pool]
    d_pool = [(fr.DeviceBuffer.from_numpy(ctx, i_), fr.DeviceBuffer.from_numpy(ctx, d_)) for i_, d_ in pool]
    outs = []
    for rep in range(64 * 2 + 9):      # two full launch groups (64-item kernel: 64 x 4 tiles > 128) and a partial one (32-item kernel)
        j, b = rep % 4, [200, 200, 77, 200, 1][rep % 5]
        buf = fr.DeviceBuffer(ctx, B * 4)
        buf.upload(np.full(B, np.nan, np.float32))
        wk.push_device(b, d_pool[j][0], d_pool[j][1], buf)
        outs.append((buf, j, b))
    wk.sync()
    tol = {"f32": 1e-5, "bf16": 5e-3, "fp8": 3e-2}[prec]
    first = {}
    for buf, j, b in outs:
        got = buf.download(np.float32, B)
        assert np.isnan(got[b:]).all()
        assert np.abs(got[:b] - expect[j][:b]).max() <= tol * np.abs(expect[j]).max(), (prec, j, b)
        if (j, b) in first:
            assert np.array_equal(got[:b], first[(j, b)])      # the 32- and 64-item kernels and every position in a group agree to the bit
        first.setdefault((j, b), got[:b].copy())
        if (j, 200) in first and b < 200:
            assert np.array_equal(got[:b], first[(j, 200)][:b])
        buf.free()
    # an out-of-range index and a NaN feature are seen through this path too
    bad = pool[0][0].copy()
    bad[199, 3] = m.rows()[3]
    d_bad = fr.DeviceBuffer.from_numpy(ctx, bad)
    d_s = fr.DeviceBuffer(ctx, B * 4)
    with pytest.raises(fr.FleetRecError) as e:
        wk.push_device(B, d_bad, d_pool[0][1], d_s)
        wk.sync()
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    dn = pool[1][1].copy()
    dn[17, 9] = np.nan
    d_dn = fr.DeviceBuffer.from_numpy(ctx, dn)
    wk.push_device(B, d_pool[1][0], d_dn, d_s)
    wk.sync()
    sc = d_s.download(np.float32, B)
    assert np.isnan(sc[17]) and np.isfinite(np.delete(sc, 17)).all()
    if prec == "bf16":
        # ... and through the persistent wave-specialised kernel (fr_fused_tile_hs_kernel<1, 22, ...>): a group of 256 batches of 200 items =
        # 1024 tiles, i.e. four per compute unit; its producers read the DENSE words from the request's feature rows.  Same bits as above.
        ctx.set_stream_group(256)
        outs = []
        for rep in range(256):
            j, b = rep % 4, [200, 200, 77, 200, 1][rep % 5]
            buf = fr.DeviceBuffer(ctx, B * 4)
            buf.upload(np.full(B, np.nan, np.float32))
            wk.push_device(b, d_pool[j][0], d_pool[j][1], buf)
            outs.append((buf, j, b))
        assert wk.last_kernel().startswith("fr_fused_tile_hs_kernel<1, 22,"), wk.last_kernel()
        wk.sync()
        for buf, j, b in outs:
            got = buf.download(np.float32, B)
            assert np.isnan(got[b:]).all() and np.array_equal(got[:b], first[(j, 200)][:b]), (j, b)
            buf.free()
        ctx.set_stream_group(64)
    wk.close()
    ctx.close()


def test_push_device_list_equals_single_pushes(fr, ctxs):
    """fr_worker_push_device_list: n pushes in one native call.  Same scores, bit for bit, as the same batches pushed one by one; an
    invalid entry stops the list at that entry with its status, the batches before it stay pushed and complete at the next sync."""
    m, ctx = ctxs(fr.MODEL_A)
    rng = np.random.default_rng(77)
    sizes = [256, 1, 200, 256, 37, 256, 255, 64] * 9     # 72 batches: one full launch group and a partial one
    idx = [uniform_idx(rng, m.rows(), b) for b in sizes[:8]]
    d_idx = [fr.DeviceBuffer.from_numpy(ctx, i) for i in idx]
    wk = fr.Worker(ctx, 256)
    one, lst = [fr.DeviceBuffer(ctx, 256 * 4) for _ in sizes], [fr.DeviceBuffer(ctx, 256 * 4) for _ in sizes]
    for b_ in one + lst:
        b_.upload(np.full(256, np.nan, np.float32))
    for j, b in enumerate(sizes):
        wk.push_device(b, d_idx[j % 8], None, one[j])
    wk.sync()
    pl = wk.make_push_list(sizes, [d_idx[j % 8] for j in range(len(sizes))], None, lst)
    wk.push_device_list(pl)
    wk.sync()
    for j, b in enumerate(sizes):
        a, c = one[j].download(np.float32, 256), lst[j].download(np.float32, 256)
        assert np.isfinite(a[:b]).all() and np.isnan(a[b:]).all() and np.array_equal(a[:b], c[:b]) and np.isnan(c[b:]).all(), j
    # an entry with a batch above the worker's capacity: the call stops there
    bad = wk.make_push_list([256, 256, 257, 256], [d_idx[0]] * 4, None, lst[:4])
    for b_ in lst[:4]:
        b_.upload(np.full(256, np.nan, np.float32))
    with pytest.raises(fr.FleetRecError) as e:
        wk.push_device_list(bad)
    assert e.value.status == fr.FR_ERR_INVALID
    wk.sync()
    got = [b_.download(np.float32, 256) for b_ in lst[:4]]
    assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all() and np.isnan(got[2]).all() and np.isnan(got[3]).all()
    wk.push_device_list(wk.make_push_list([], [], None, []))   # n = 0: nothing
    wk.close()
    for b_ in one + lst + d_idx:
        b_.free()


def test_host_fed_streaming(fr, ctxs):
    """fr_worker_push_host / fr_driver_run_host_streaming: batches that sit in host memory are staged in pinned blocks and travel as
    one H2D + one fused launch + one D2H per block; scores equal the device-resident streaming path bit for bit, ragged batches,
    partial blocks, several trips round the 4 staging blocks."""
    m, ctx = ctxs(fr.MODEL_A)
    rng = np.random.default_rng(77)
    sizes = [256, 1, 200, 256, 37] * 70      # 350 pushes: > 4 blocks of 64
    pool = [uniform_idx(rng, m.rows(), 256) for _ in range(5)]
    wk = fr.Worker(ctx, 256)
    d_pool = [fr.DeviceBuffer.from_numpy(ctx, p_) for p_ in pool]
    d_sc = [fr.DeviceBuffer(ctx, 256 * 4) for _ in range(5)]
    for j in range(5):
        wk.push_device(sizes[j], d_pool[j], None, d_sc[j])
    wk.sync()
    expect = [d_sc[j].download(np.float32, 256)[:sizes[j]] for j in range(5)]
    outs = [np.full(256, np.nan, np.float32) for _ in sizes]
    for j, b in enumerate(sizes):
        wk.push_host(pool[j % 5][:b], None, outs[j])
        if j == 100:
            wk.sync()                         # a sync in the middle of a block
    wk.sync()
    for j, b in enumerate(sizes):
        assert np.array_equal(outs[j][:b], expect[j % 5]), j
        assert np.isnan(outs[j][b:]).all()
    # the zero-copy form: the caller writes into the worker's pinned staging slot (fr_worker_stage_acquire) and queues it
    # (fr_worker_push_staged), interleaved with copying pushes; same bits
    outs2 = [np.full(256, np.nan, np.float32) for _ in sizes]
    for j, b in enumerate(sizes):
        if j % 3 == 2:
            wk.push_host(pool[j % 5][:b], None, outs2[j])
        else:
            slot, dslot = wk.stage_acquire(256 if j % 2 else b)     # a slot may be acquired larger than what is pushed
            assert dslot is None and slot.shape[1] == m.idx_cols
            slot[:b] = pool[j % 5][:b]
            if j == 7:
                with pytest.raises(fr.FleetRecError) as e:          # one slot at a time; no copying push in between
                    wk.push_host(pool[0], None, outs2[j])
                assert e.value.status == fr.FR_ERR_STATE
                with pytest.raises(fr.FleetRecError) as e:
                    wk.stage_acquire(256)
                assert e.value.status == fr.FR_ERR_STATE
            wk.push_staged(b, outs2[j])
    with pytest.raises(fr.FleetRecError) as e:
        wk.push_staged(1, outs2[0])                                 # nothing acquired
    assert e.value.status == fr.FR_ERR_STATE
    wk.stage_acquire(256)                                           # acquired and never pushed: dropped by sync
    wk.sync()
    for j, b in enumerate(sizes):
        assert np.array_equal(outs2[j][:b], expect[j % 5]), j
        assert np.isnan(outs2[j][b:]).all()
    # serving with replies: fr_worker_flush launches a partial block without waiting, fr_worker_host_poll delivers finished blocks (in
    # push order) and counts them -- no fr_worker_sync anywhere in this stretch
    import time
    base = wk.host_poll()
    outs3 = [np.full(256, np.nan, np.float32) for _ in range(10)]
    for j in range(10):
        wk.push_host(pool[j % 5], None, outs3[j])
    assert wk.host_poll() == base          # 10 batches do not fill a block of 64: nothing has been launched
    wk.flush()
    t0 = time.time()
    while wk.host_poll() < base + 10:
        assert time.time() - t0 < 30
        time.sleep(0.0005)
    for j in range(10):
        assert np.array_equal(outs3[j], expect[j % 5]) if sizes[j % 5] == 256 else True
        assert np.array_equal(outs3[j][:sizes[j % 5]], expect[j % 5])
    wk.flush()                             # nothing queued: a no-op
    assert wk.host_poll() == base + 10
    assert wk.host_pending() == (0, 0, 0)
    # fr_ctx_set_small_block: a block that leaves with <= 2 batches takes fr_worker_submit's stage launches (latency of a nearly idle
    # server) and gets submit's scores bit for bit; a bigger block still takes the fused kernel
    wk.sync()                              # (fr_worker_submit wants an idle worker: the polls above delivered everything but did not sync)
    sub = [wk.infer(p_) for p_ in pool[:2]]
    ctx.set_small_block(2)
    o1 = [np.full(256, np.nan, np.float32) for _ in range(5)]
    wk.push_host(pool[0], None, o1[0])
    wk.push_host(pool[1], None, o1[1])
    assert wk.host_pending()[0] == 2
    wk.flush()
    for j in (2, 3, 4):
        wk.push_host(pool[j][:sizes[j]], None, o1[j])
    wk.flush()
    t0 = time.time()
    while wk.host_poll() < base + 15:
        assert time.time() - t0 < 30
        time.sleep(0.0005)
    assert np.array_equal(o1[0], sub[0]) and np.array_equal(o1[1], sub[1])
    for j in (2, 3, 4):
        assert np.array_equal(o1[j][:sizes[j]], expect[j])
    ctx.set_small_block(0)
    with pytest.raises(fr.FleetRecError):
        ctx.set_small_block(9)
    bad = pool[0].copy()
    bad[3, 5] = m.rows()[5]
    wk.push_host(bad, None, outs[0])
    with pytest.raises(fr.FleetRecError) as e:
        wk.sync()
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    wk.close()
    # the native loop
    drv = fr.Driver(ctx, 2, 2, 256)
    el = drv.run_host(256, 3000, pool, streaming=True)
    assert el > 0
    w0 = fr.Worker(ctx, 256)
    full = [w0.infer(p_) for p_ in pool]
    w0.close()
    scale = max(np.abs(f).max() for f in full)
    seen = 0
    for t in range(2):
        for sl in range(2):
            for row in drv.host_score_ring(t, sl, 256):
                if row.any():
                    assert min(np.abs(row - f).max() for f in full) <= 1e-5 * scale
                    seen += 1
    assert seen >= 4 * 128
    drv.close()
    # a model that does not stream through the fused kernel is refused
    mc, cc = ctxs(fr.MODEL_C)
    wc = fr.Worker(cc, 64)
    with pytest.raises(fr.FleetRecError) as e:
        wc.push_host(uniform_idx(rng, mc.rows(), 64), np.zeros((64, mc.dense_len), np.float32), np.zeros(64, np.float32))
    assert e.value.status == fr.FR_ERR_STATE
    wc.close()


@pytest.mark.parametrize("prec,seed", [("f32", 1), ("f32", 2), ("bf16", 3)])
def test_random_streaming_sequences(fr, gpu, prec, seed):
    """State machine of the streaming entry points under a random schedule: device pushes, copying host pushes, zero-copy staged
    pushes, syncs and changes of the launch group in any order, ragged batches.  Every batch's scores must come out bit-identical to the
    first (all-device, one group size) run of the same index rows -- whatever was queued around it."""
    m = fr.Model.builtin(fr.MODEL_A).clone(max_rows=20000)
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    ctx.set_fc_precision({"f32": fr.FC_FP32, "bf16": fr.FC_BF16}[prec])
    rng = np.random.default_rng(900 + seed)
    pool = [uniform_idx(rng, m.rows(), 256) for _ in range(6)]
    d_pool = [fr.DeviceBuffer.from_numpy(ctx, p_) for p_ in pool]
    wk = fr.Worker(ctx, 256)
    ref_buf = [fr.DeviceBuffer(ctx, 256 * 4) for _ in pool]
    for j in range(len(pool)):
        wk.push_device(256, d_pool[j], None, ref_buf[j])
    wk.sync()
    ref = [b.download(np.float32, 256) for b in ref_buf]
    # device pushes under a launch group below 12 ride the stage pipeline: the unpipelined submit's bits for the same batch size (same split-K plan)
    ref_pipe = {(j, b): wk.infer(pool[j][:b]).copy() for j in range(len(pool)) for b in (1, 7, 64, 200, 256)}
    n_ops = 260
    dev_out, host_out, plan = [], [], []
    g0 = ctx.stream_group()
    for op in range(n_ops):
        r = rng.random()
        j, b = int(rng.integers(0, len(pool))), int(rng.choice([1, 7, 64, 200, 256]))
        if r < 0.35:
            buf = fr.DeviceBuffer(ctx, 256 * 4)
            buf.upload(np.full(256, np.nan, np.float32))
            wk.push_device(b, d_pool[j], None, buf)
            dev_out.append((buf, j, b, ctx.stream_group() < 12))
        elif r < 0.60:
            out = np.full(256, np.nan, np.float32)
            wk.push_host(pool[j][:b], None, out)
            host_out.append((out, j, b))
        elif r < 0.85:
            out = np.full(256, np.nan, np.float32)
            slot, _ = wk.stage_acquire(b)
            slot[:b] = pool[j][:b]
            wk.push_staged(b, out)
            host_out.append((out, j, b))
        elif r < 0.93:
            wk.sync()
        else:
            ctx.set_stream_group(int(rng.choice([1, 3, 16, 64])))
    wk.sync()
    ctx.set_stream_group(g0)
    for buf, j, b, piped in dev_out:
        got = buf.download(np.float32, 256)
        assert np.array_equal(got[:b], ref_pipe[(j, b)] if piped else ref[j][:b]) and np.isnan(got[b:]).all(), (j, b, piped)
        buf.free()
    for out, j, b in host_out:
        assert np.array_equal(out[:b], ref[j][:b]) and np.isnan(out[b:]).all()
    assert len(dev_out) > 50 and len(host_out) > 80
    wk.close()
    ctx.close()


@pytest.mark.parametrize("prec", ["f32", "bf16", "fp8"])
def test_random_stage_pipeline_sequences(fr, gpu, prec):
    """The same for a model that streams through the STAGE pipeline (Model-C: launch L = gather(L) | FC1(L-1) | ... | out(L-4), two
    activation sets alternating by launch parity): random ragged batches and syncs; every pushed batch equals the unpipelined
    submit() of the same rows bit for bit (same stage bodies, same split-K plan for the same batch size)."""
    m = fr.Model.builtin(fr.MODEL_C).clone(max_rows=5000)
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    ctx.set_fc_precision({"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[prec])
    rng = np.random.default_rng(4242)
    MAXB = 192
    pool = [uniform_idx(rng, m.rows(), MAXB) for _ in range(4)]
    dpool = [rng.uniform(-1, 1, (MAXB, m.dense_len)).astype(np.float32) for _ in range(4)]
    wk = fr.Worker(ctx, MAXB)
    if prec == "fp8":
        wk.calibrate_fp8(pool[0], dpool[0])
    sizes = [1, 33, 64, 100, 192]
    ref = {(j, b): wk.infer(pool[j][:b], dpool[j][:b]).copy() for j in range(4) for b in sizes}
    d_idx = {(j, b): fr.DeviceBuffer.from_numpy(ctx, pool[j][:b]) for j in range(4) for b in sizes}
    d_dense = {(j, b): fr.DeviceBuffer.from_numpy(ctx, dpool[j][:b]) for j in range(4) for b in sizes}
    outs = []
    for op in range(120):
        if rng.random() < 0.12:
            wk.sync()
            continue
        j, b = int(rng.integers(0, 4)), int(rng.choice(sizes))
        buf = fr.DeviceBuffer(ctx, MAXB * 4)
        buf.upload(np.full(MAXB, np.nan, np.float32))
        wk.push_device(b, d_idx[(j, b)], d_dense[(j, b)], buf)
        outs.append((buf, j, b))
    wk.sync()
    for buf, j, b in outs:
        got = buf.download(np.float32, MAXB)
        assert np.array_equal(got[:b], ref[(j, b)]), (prec, j, b)
        assert np.isnan(got[b:]).all()
        buf.free()
    assert len(outs) > 80
    wk.close()
    ctx.close()


@pytest.mark.gpu
def test_launch_groups_above_64(fr, O, ctxs):
    """fr_ctx_set_stream_group above 64 (round 3): the persistent bf16 kernel takes its batch list from device memory, so ONE launch carries
    up to 256 batches -- Model-A's batches of 256 items reach "two tiles per compute unit" at a group of 128+ and then stream through
    fr_fused_tile_hs_kernel<1, 22, ...>; every other kernel carries at most 64 batches per launch and a larger group leaves in slices of 64.
    Scores are the same bits whichever group size / kernel carried a batch; 257 is refused."""
    m, ctx = ctxs(fr.MODEL_A)
    rng = np.random.default_rng(91)
    B = 256
    pool = [uniform_idx(rng, m.rows(), B) for _ in range(5)]
    d_pool = [fr.DeviceBuffer.from_numpy(ctx, i_) for i_ in pool]
    g0 = ctx.stream_group()
    try:
        for prec, enum, kernel_big in (("bf16", fr.FC_BF16, "fr_fused_tile_hs_kernel<1, 22,"), ("f32", fr.FC_FP32, "fr_fused_tile_")):   # (fp32 launches at 16384 queued items: slices of <= 64 batches through the kernarg-fed kernels)
            ctx.set_fc_precision(enum)
            wk = fr.Worker(ctx, B)
            ctx.set_stream_group(64)
            base = []
            for j in range(5):
                o = fr.DeviceBuffer(ctx, B * 4)
                wk.push_device(B, d_pool[j], None, o)
                base.append(o)
            wk.sync()
            small_kernel = wk.last_kernel()
            want = [o.download(np.float32, B) for o in base]
            ctx.set_stream_group(256)
            assert ctx.stream_group() == 256
            outs = [fr.DeviceBuffer(ctx, B * 4) for _ in range(256 + 37)]
            sizes = [256, 256, 200, 256, 1, 256, 64]
            for i, o in enumerate(outs):
                o.upload(np.full(B, np.nan, np.float32))
                wk.push_device(sizes[i % len(sizes)], d_pool[i % 5], None, o)
                if i == 255:
                    assert wk.last_kernel().startswith(kernel_big), (prec, wk.last_kernel())   # the full group of 256 has just been launched
            wk.sync()
            assert wk.last_kernel() != "" and (prec == "f32" or wk.last_kernel() == small_kernel or wk.last_kernel().startswith("fr_fused_tile_h"))
            for i, o in enumerate(outs):
                b = sizes[i % len(sizes)]
                got = o.download(np.float32, B)
                assert np.isnan(got[b:]).all() and np.array_equal(got[:b], want[i % 5][:b]), (prec, i, b)
                o.free()
            for o in base:
                o.free()
            wk.close()
        with pytest.raises(fr.FleetRecError):
            ctx.set_stream_group(257)
        # Model-B, batches of 1024: a group of 128 = 131 072 items = 2 048 tiles in ONE launch (8 per persistent workgroup) -- more than the
        # 65 536 items a bf16 launch carried before the item cap followed the group.  Same bits as the chunked kernel's small launch.
        mb, cb = ctxs(fr.MODEL_B)
        gb = cb.stream_group()
        cb.set_fc_precision(fr.FC_BF16)
        try:
            poolb = [fr.DeviceBuffer.from_numpy(cb, uniform_idx(rng, mb.rows(), 1024)) for _ in range(3)]
            wkb = fr.Worker(cb, 1024)
            cb.set_stream_group(64)
            baseb = [fr.DeviceBuffer(cb, 1024 * 4) for _ in range(3)]
            for j in range(3):
                wkb.push_device(1024, poolb[j], None, baseb[j])
            wkb.sync()
            assert wkb.last_kernel().startswith("fr_fused_tile_h_kernel<"), wkb.last_kernel()
            wantb = [o.download(np.float32, 1024) for o in baseb]
            cb.set_stream_group(128)
            outb = [fr.DeviceBuffer(cb, 1024 * 4) for _ in range(128)]
            szb = [1024, 1024, 1000, 1024, 513]
            for i, o in enumerate(outb):
                o.upload(np.full(1024, np.nan, np.float32))
                wkb.push_device(szb[i % 5], poolb[i % 3], None, o)
                assert wkb.last_kernel().startswith("fr_fused_tile_h_kernel<") or i == 127, (i, wkb.last_kernel())   # nothing leaves before the 128th push
            assert wkb.last_kernel().startswith("fr_fused_tile_hs_kernel<1, 55,"), wkb.last_kernel()
            wkb.sync()
            for i, o in enumerate(outb):
                b = szb[i % 5]
                got = o.download(np.float32, 1024)
                assert np.isnan(got[b:]).all() and np.array_equal(got[:b], wantb[i % 3][:b]), (i, b)
                o.free()
            for o in baseb + poolb:
                o.free()
            wkb.close()
        finally:
            cb.set_stream_group(gb)
            cb.set_fc_precision(fr.FC_FP32)
    finally:
        ctx.set_stream_group(g0)
        ctx.set_fc_precision(fr.FC_FP32)
        for d in d_pool:
            d.free()
