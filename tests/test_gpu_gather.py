"""GPU parity, part 1: table contents and the gather (records bit-exact against the oracle in every index mode, layout and kernel variant).
Tolerances, seeds and reference chains: tests/gpu_helpers.py; the full-size contexts (`ctxs`): tests/conftest.py."""
import os
import threading

import numpy as np
import pytest
from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_fill_kernels_match_oracle_content(fr, O, ctxs):
    """Device-side table synthesis == the oracle's content function (all three modes, several tables)."""
    m, ctx = ctxs(fr.MODEL_B)
    tabs = m.tables()
    picks = [0, 16, 41, 50, len(tabs) - 1]
    for mode, omode in ((fr.FILL_HASH, O.FILL_HASH), (fr.FILL_TAGGED, O.FILL_TAGGED), (fr.FILL_EVEN_ODD, O.FILL_EVEN_ODD)):
        ctx.fill_tables(mode, 1234)
        for t in picks:
            d = tabs[t]
            uid = d.source * 1024 + d.mem_class * 256 + d.table_id
            for row0 in (0, max(0, d.rows - 300)):
                n = min(300, d.rows - row0)
                got = ctx.download_table(t, row0, n)
                assert np.array_equal(got, O.content_rows(omode, 1234, uid, n, d.dim, row0=row0)), (mode, t, row0)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)


@pytest.mark.parametrize("which,B", [(0, 256), (1, 1024), (2, 4096)])
def test_gather_bit_exact_full_models(fr, O, ctxs, which, B):
    """BASELINE configs 2/3/4 shapes: uniform random per-table indices over the FULL row ranges."""
    m, ctx = ctxs(which)
    om = O.OracleModel(NAMES[which])
    assert np.array_equal(m.rows(), om.rows_wire)
    rng = np.random.default_rng(1234)
    idx = uniform_idx(rng, m.rows(), B)
    idx[0] = 0
    idx[1] = m.rows() - 1  # maximum index of every table
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    wk = fr.Worker(ctx, B)
    got = wk.gather_records(idx, dense).reshape(B, m.record_len)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    assert np.array_equal(got, want)
    wk.close()


@pytest.mark.parametrize("which", [0, 1, 2])
def test_gather_tagged_wire_order(fr, O, gpu, which):
    """Tagged tables: every float of every record names its (source, class, table, row, col)."""
    m = fr.Model.builtin(which).clone(max_rows=5000)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_TAGGED, 0)
    rng = np.random.default_rng(5)
    B = 37  # ragged: not a multiple of the kernel's items-per-block
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    wk = fr.Worker(ctx, 64)
    got = wk.gather_records(idx, dense).reshape(B, m.record_len)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_TAGGED)
    assert np.array_equal(got, want)
    # batch = 1 (BASELINE config 1 shape) and the empty-ish edge
    got1 = wk.gather_records(idx[:1], None if dense is None else dense[:1])
    assert np.array_equal(got1, want[0])
    wk.close()
    ctx.close()


@pytest.mark.parametrize("mode", ["table", "bank", "item"])
def test_wide_record_gather_ragged_batches_and_transports(fr, O, gpu, mode):
    """gather_pack_stream_kernel (records of >= 512 words at batch >= 1024: the software-pipelined, XCD-partitioned form whose index
    loads and record stores are bounded by buffer resources instead of branches): batches that end inside a chunk, inside a
    workgroup's second chunk and exactly on one, in every index mode; bit-exact against the oracle.  The bf16 / e4m3 transport forms
    of the same launch against RNE of the fp32 records and against the narrow-batch kernel (batch < 1024 takes gather_pack_kernel)
    on the same items; an out-of-range index in the ragged tail is reported."""
    imode = {"table": fr.INDEX_PER_TABLE, "bank": fr.INDEX_PER_BANK, "item": fr.INDEX_PER_ITEM}[mode]
    m = fr.Model.builtin(fr.MODEL_C).clone(max_rows=30000, index_mode=imode)
    om = O.OracleModel("C")
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    rng = np.random.default_rng(4242)
    BMAX = 2056
    wk = fr.Worker(ctx, BMAX)
    dense = rng.uniform(-1, 1, (BMAX, m.dense_len)).astype(np.float32)
    if mode == "table":
        idx = uniform_idx(rng, m.rows(), BMAX)
        want = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    elif mode == "bank":
        _, brows = m.bank_map()
        idx = uniform_idx(rng, brows, BMAX)
        want = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES, per_bank=True)
    else:
        idx = rng.integers(0, int(m.rows().min()), (BMAX, 1), dtype=np.int32)
        want = om.gather(np.repeat(idx, m.n_tables, axis=1), dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES)
    for B in (1024, 1027, 1029, 2050, BMAX):
        got = wk.gather_records(idx[:B], dense[:B]).reshape(B, m.record_len)
        assert np.array_equal(got, want[:B]), (mode, B)
    # the one-chunk-per-workgroup kernel (what records or index buffers of >= 4000 MiB fall back to), selected explicitly
    ctx.set_gather_variant(fr.GATHER_WORD_MAJOR_ONE_CHUNK)
    try:
        assert np.array_equal(wk.gather_records(idx[:1027], dense[:1027]).reshape(1027, m.record_len), want[:1027]), mode
    finally:
        ctx.set_gather_variant(fr.GATHER_WORD_MAJOR)
    f32 = want.view(np.float32)
    d_i = fr.DeviceBuffer.from_numpy(ctx, idx)
    d_d = fr.DeviceBuffer.from_numpy(ctx, dense)
    for B in (1027, BMAX):
        d_sl = fr.DeviceBuffer(ctx, B * m.record_len * 2)
        wk.gather_slices(B, d_i, d_d, d_sl, fr.FC_BF16)
        wk.sync()
        got16 = d_sl.download(np.uint16, B * m.record_len).reshape(B, m.record_len)
        assert np.array_equal(got16, (bf16_round(f32[:B]).view(np.uint32) >> 16).astype(np.uint16)), (mode, B)
        d_sl.free()
    # an out-of-range index in the ragged tail is seen (and the flag is cleared once reported)
    bad = idx[:1027].copy()
    bad[1026, -1] = 2 ** 30
    d_bad = fr.DeviceBuffer.from_numpy(ctx, bad)
    d_rec = fr.DeviceBuffer(ctx, 1027 * m.record_len * 4)
    with pytest.raises(fr.FleetRecError) as e:
        wk.gather_only(1027, d_bad, d_d, d_rec)
        wk.sync()
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    wk.gather_only(1027, d_i, d_d, d_rec)
    wk.sync()
    assert np.array_equal(d_rec.download(np.uint32, 1027 * m.record_len).reshape(1027, -1), want[:1027])
    wk.close()
    # e4m3 transport: the same bytes as the narrow-batch kernel writes for the same items
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    ctx.set_fc_precision(fr.FC_FP8)
    wk8 = fr.Worker(ctx, BMAX)
    wk8.calibrate_fp8(idx[:256], dense[:256])
    d_a, d_b = fr.DeviceBuffer(ctx, BMAX * m.record_len), fr.DeviceBuffer(ctx, 1000 * m.record_len)
    wk8.gather_slices(BMAX, d_i, d_d, d_a, fr.FC_FP8)
    wk8.sync()
    wk8.gather_slices(1000, d_i, d_d, d_b, fr.FC_FP8)
    wk8.sync()
    a8 = d_a.download(np.uint8, BMAX * m.record_len).reshape(BMAX, m.record_len)
    assert np.array_equal(a8[:1000], d_b.download(np.uint8, 1000 * m.record_len).reshape(1000, m.record_len))
    act_exp, _ = ctx.fp8_exponents()
    assert np.array_equal(a8, e4m3_encode(f32 * np.float32(2.0 ** act_exp[0])))
    wk8.close()
    ctx.close()


def test_wide_shard_slices_keep_their_padding(fr, gpu):
    """The software-pipelined gather on SHARD slices of >= 512 words whose lengths differ: the slice buffer is [batch][padded slice], wider
    than the shard's own words, and the kernel's store resource must be bounded by that buffer, not by the words the shard writes (a
    bound of batch x own words would silently drop the last items' stores).  Two shards of a 76-table user model, batch >= 1024,
    both slices against the segment-by-segment definition; the pad columns of the shorter slice stay untouched."""
    rng = np.random.default_rng(99)
    dims = [64] * 33 + [32] * 2 + [64] * 41          # 4800 floats; the float-balanced cut leaves two slices of different length
    tabs = [{"dim": d_, "rows": int(rng.integers(30, 3000))} for d_ in dims]
    m = fr.Model.from_spec({"name": "wide_shards", "tables": tabs, "fc": [1024, 512, 256]})
    offs, lens, F = m.shard_plan(2)
    assert min(lens) // 4 >= 512 and lens[0] != lens[1] and F == max(lens)
    host = [rng.standard_normal((t["rows"], t["dim"])).astype(np.float32) for t in tabs]
    B = 1024 + 37
    idx = uniform_idx(rng, m.rows(), B)
    want = np.empty((B, m.record_len), np.float32)
    for sg in m.segments():
        want[:, sg.rec_offset:sg.rec_offset + sg.len] = host[sg.src][idx[:, sg.src], sg.src_col:sg.src_col + sg.len]
    for r in range(2):
        ctx = fr.Context(m, device=gpu, shard_rank=r, n_shards=2)
        for sg in m.segments():
            if offs[r] <= sg.rec_offset < offs[r] + lens[r]:
                ctx.upload_table(sg.src, host[sg.src])
        wk = fr.Worker(ctx, B)
        d_i = fr.DeviceBuffer.from_numpy(ctx, idx)
        d_sl = fr.DeviceBuffer(ctx, B * F * 4)
        d_sl.upload(np.full(B * F, 0x7fc01234, np.uint32))
        wk.gather_only(B, d_i, None, d_sl)
        wk.sync()
        sl = d_sl.download(np.uint32, B * F).reshape(B, F)
        assert np.array_equal(sl[:, :lens[r]], want[:, offs[r]:offs[r] + lens[r]].view(np.uint32)), r
        assert (sl[:, lens[r]:] == 0x7fc01234).all()
        wk.close()
        ctx.close()


def test_wide_record_gather_with_a_dense_block(fr, gpu):
    """The same kernel on a user-defined model whose 648-word record carries a dense block between two table sources (dense words
    take their item number, not an index): ragged batch >= 1024, against the segment-by-segment definition of the record."""
    rng = np.random.default_rng(77)
    tabs = [{"dim": int(rng.choice([16, 32, 64])), "rows": int(rng.integers(50, 5000))} for _ in range(72)]
    m = fr.Model.from_spec({"name": "wide_dense", "tables": tabs, "dense_len": 24, "dense_at": 31, "fc": [1024, 512, 256]})
    assert m.record_len // 4 >= 512
    ctx = fr.Context(m, device=gpu)
    host = [rng.standard_normal((t["rows"], t["dim"])).astype(np.float32) for t in tabs]
    for t, a in enumerate(host):
        ctx.upload_table(t, a)
    B = 1024 + 203
    idx = uniform_idx(rng, m.rows(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    want = np.empty((B, m.record_len), np.float32)
    for sg in m.segments():
        if sg.kind == fr.SEG_DENSE:
            want[:, sg.rec_offset:sg.rec_offset + sg.len] = dense[:, sg.src_col:sg.src_col + sg.len]
        else:
            want[:, sg.rec_offset:sg.rec_offset + sg.len] = host[sg.src][idx[:, sg.src], sg.src_col:sg.src_col + sg.len]
    wk = fr.Worker(ctx, B)
    assert np.array_equal(wk.gather_records(idx, dense).reshape(B, m.record_len), want.view(np.uint32))
    wk.close()
    ctx.close()


@pytest.mark.parametrize("which,mode", [(0, "table"), (1, "table"), (2, "table"), (2, "bank"), (1, "item")])
def test_gather_kernel_variants_are_bit_identical(fr, O, gpu, which, mode):
    """fr_ctx_set_gather_variant: the item-tile gather (LDS-staged row packing) with and without the wave-level merge of duplicate
    lookups (LDS hash + __shfl, counted with __ballot) writes the same records as the word-major kernel and as the oracle -- ragged
    batch, indices with many repeats (so that leaders, duplicates and hash-slot collisions all occur), every index mode."""
    imode = {"table": fr.INDEX_PER_TABLE, "bank": fr.INDEX_PER_BANK, "item": fr.INDEX_PER_ITEM}[mode]
    m = fr.Model.builtin(which).clone(max_rows=50000, index_mode=imode)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_TAGGED, 0)
    rng = np.random.default_rng(77)
    B = 333
    ranges = m.index_ranges()
    idx = uniform_idx(rng, np.minimum(ranges, 40), B)        # <= 40 distinct rows per column: most lookups repeat inside a wave
    idx[::7] = uniform_idx(rng, ranges, len(idx[::7]))       # ... plus full-range rows
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    if mode == "table":
        want = om.gather(idx, dense=dense, content_mode=O.FILL_TAGGED)
    elif mode == "bank":
        want = om.gather(idx, dense=dense, content_mode=O.FILL_TAGGED, per_bank=True)
    else:
        want = om.gather(idx[:, 0], dense=dense, content_mode=O.FILL_TAGGED)
    wk = fr.Worker(ctx, B)
    for var in (fr.GATHER_WORD_MAJOR, fr.GATHER_ITEM_TILE, fr.GATHER_ITEM_TILE_DEDUP, fr.GATHER_ITEM_TILE_DEDUP_COUNT):
        ctx.set_gather_variant(var)
        got = wk.gather_records(idx, dense).reshape(B, m.record_len)
        assert np.array_equal(got, want), var
    merged = ctx.gather_merged_lookups()
    lookups = B * (m.n_tables if mode != "bank" else m.idx_cols)
    # the waves did merge repeated rows (a bank row is cut in power-of-two pieces, each piece counts its own merges: <= 3 per lookup)
    assert 0.3 * lookups < merged <= 3 * lookups, (merged, lookups)
    # out-of-range indices are still reported by the item-tile kernels
    bad = idx.copy()
    bad[5, 0] = int(ranges[0])
    d_idx = fr.DeviceBuffer.from_numpy(ctx, bad)
    d_dense = fr.DeviceBuffer.from_numpy(ctx, dense) if dense is not None else None
    d_rec = fr.DeviceBuffer(ctx, B * m.record_len * 4)
    wk.gather_only(B, d_idx, d_dense, d_rec)
    with pytest.raises(fr.FleetRecError) as e:
        wk.sync()
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    wk.close()
    ctx.close()


@pytest.mark.parametrize("which,mode", [(2, "table"), (2, "bank"), (1, "bank"), (1, "table")])
def test_gather_groups_cut_on_source_rows(fr, O, gpu, which, mode):
    """fr_ctx_gather_groups: the 8 word ranges the word-major gather deals to the XCDs cover the record exactly once, are at most
    256 words wide, and every cut sits between two SOURCE rows -- a table row (segment) with per-table indices, a whole bank row
    (the consecutive segments of one bank) with per-bank indices -- so that no row is fetched through two L2s; and the records of a
    batch large enough to take that kernel (>= 1024 items) are still bit-exact against the oracle."""
    imode = {"table": fr.INDEX_PER_TABLE, "bank": fr.INDEX_PER_BANK}[mode]
    m = fr.Model.builtin(which).clone(max_rows=30000, index_mode=imode)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_TAGGED, 0)
    st = ctx.gather_groups()
    n_words = m.record_len // 4
    assert st[0] == 0 and st[8] == n_words and all(0 < st[g + 1] - st[g] <= 256 for g in range(8)), st
    assert max(st[g + 1] - st[g] for g in range(8)) <= 1.15 * n_words / 8 + 16, st      # still balanced
    bot, _ = m.bank_map()
    allowed = {0, n_words}
    prev = None
    for sg in m.segments():
        key = ("dense",) if sg.kind == fr.SEG_DENSE else (("bank", int(bot[sg.src])) if mode == "bank" and sg.kind == fr.SEG_TABLE else ("seg", sg.rec_offset))
        if key != prev:
            allowed.add(sg.rec_offset // 4)
        if sg.kind == fr.SEG_DENSE:
            allowed.update(range(sg.rec_offset // 4, (sg.rec_offset + sg.len) // 4 + 1, 8))   # the dense block may be cut every 128 bytes
        prev = key
    assert set(st) <= allowed, (st, sorted(set(st) - allowed))
    B = 1024 + 37
    rng = np.random.default_rng(99)
    idx = uniform_idx(rng, m.index_ranges(), B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    want = om.gather(idx, dense=dense, content_mode=O.FILL_TAGGED, per_bank=(mode == "bank"))
    wk = fr.Worker(ctx, B)
    assert np.array_equal(wk.gather_records(idx, dense).reshape(B, m.record_len), want)
    wk.close()
    ctx.close()


@pytest.mark.parametrize("which,fname", [(0, "records_47.bin"), (1, "records_98.bin"), (2, "records_377x2.bin")])
def test_gather_matches_committed_golden_records(fr, gpu, which, fname):
    """The device gather against COMMITTED bytes (tests/golden/records_*.bin, tagged tables: every float names its table / row /
    column), not only against the live oracle."""
    import importlib.util
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_records", os.path.join(gold, "make_records.py"))
    # only the reader of the fixture format is used here; the module imports the oracle to be able to WRITE fixtures
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    idx, dense, rec = mk.read(os.path.join(gold, fname))
    m = fr.Model.builtin(which).clone(max_rows=60000)
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_TAGGED, 0)
    wk = fr.Worker(ctx, 32)
    got = wk.gather_records(idx, dense if m.dense_len else None).reshape(32, m.record_len)
    assert np.array_equal(got, rec)
    # the fused / pipeline kernels' own gather stage (feature-major activations of a submit) carries the same bits
    ctx.fill_weights(fr.WEIGHTS_ONES, 0)
    wk.infer(idx, dense if m.dense_len else None)
    assert np.array_equal(wk.features(32), rec.T)
    wk.close()
    ctx.close()


@pytest.mark.parametrize("which,B", [(0, 256), (1, 1024), (2, 4096)])
def test_per_bank_gather_bit_exact_full_models(fr, O, gpu, ctxs, which, B):
    """FR_INDEX_PER_BANK = the kernel's real contract (ONE index per bank per item, reused by every round of the bank:
    embedding_98_krnl.cpp:1026-1040, embedding_377_krnl.cpp:1261-1290) on the bank-interleaved HBM layout, full-size tables:
    records bit-exact against the oracle's per-bank mode (bank memories addressed at ADDR_AXI + idx*AXI_PADDED_SIZE), uniform
    indices over every bank's whole valid range incl. 0 and the maximum; and scores bit-identical to the PER_TABLE context fed
    the same index expanded per table (same kernels, same arithmetic, different table layout)."""
    m = fr.Model.builtin(which).clone(index_mode=fr.INDEX_PER_BANK)   # own context, closed at the end (HBM budget of the module)
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    om = O.OracleModel(NAMES[which])
    bot, brows = m.bank_map()
    assert m.idx_cols == om.n_banks and np.array_equal(brows, om.bank_rows_wire())
    rng = np.random.default_rng(4321)
    idx = uniform_idx(rng, brows, B)
    idx[0] = 0
    idx[1] = brows - 1
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    wk = fr.Worker(ctx, B)
    got = wk.gather_records(idx, dense).reshape(B, m.record_len)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_HASH, seed=SEED_TABLES, per_bank=True)
    assert np.array_equal(got, want)
    # the same request as per-table indices on the classic layout
    mt, ctx_t = ctxs(which)
    wt = fr.Worker(ctx_t, B)
    idx_t = idx[:, bot]
    assert np.array_equal(wt.gather_records(idx_t, dense).reshape(B, m.record_len), want)
    s_bank, s_tab = wk.infer(idx, dense), wt.infer(idx_t, dense)
    assert np.array_equal(s_bank, s_tab)
    ws = [ctx.get_weights(l) for l in range(4)]
    n_chk = min(B, 512)
    ref = om.fc_chain(want[:n_chk].view(np.float32), ws, acc64=True)
    assert rel_err(s_bank[:n_chk], ref) <= 1e-3
    # streaming path (fused item-tile kernels for A / B, stage pipeline for C)
    d_idx = fr.DeviceBuffer.from_numpy(ctx, idx)
    d_dense = fr.DeviceBuffer.from_numpy(ctx, dense) if dense is not None else None
    d_sc = fr.DeviceBuffer(ctx, B * 4)
    wk.push_device(B, d_idx, d_dense, d_sc)
    wk.sync()
    pushed = d_sc.download(np.float32, B)
    assert rel_err(pushed, s_bank) <= 1e-5
    wk.close()
    wt.close()
    ctx.close()


@pytest.mark.parametrize("which", [1, 2])
def test_per_bank_tagged_upload_and_range(fr, O, gpu, which):
    """Bank-interleaved layout plumbing on row-capped models: tagged records (every float names table/row/col) against the
    oracle's per-bank mode at a ragged batch; upload / download of rows on both sides of the interleaved region's end; an index
    that one table of the bank cannot serve is reported (the reference would read the next table, embedding_47_krnl.cpp:927-933)."""
    m = fr.Model.builtin(which).clone(max_rows=3000, index_mode=fr.INDEX_PER_BANK)
    om = O.OracleModel(NAMES[which])
    ctx = fr.Context(m, device=gpu)
    ctx.fill_tables(fr.FILL_TAGGED, 0)
    bot, brows = m.bank_map()
    rng = np.random.default_rng(11)
    B = 45
    idx = uniform_idx(rng, brows, B)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) if m.dense_len else None
    wk = fr.Worker(ctx, 64)
    got = wk.gather_records(idx, dense).reshape(B, m.record_len)
    want = om.gather(idx, dense=dense, content_mode=O.FILL_TAGGED, per_bank=True)
    assert np.array_equal(got, want)
    # a table whose bank holds a smaller one: its rows straddle the end of the interleaved region
    tabs = m.tables()
    t = next(t for t in range(m.n_tables) if tabs[t].rows > brows[bot[t]] + 40)
    cut = int(brows[bot[t]])
    d = tabs[t]
    uid = d.source * 1024 + d.mem_class * 256 + d.table_id
    assert np.array_equal(ctx.download_table(t, cut - 20, 60), O.content_rows(O.FILL_TAGGED, 0, uid, 60, d.dim, row0=cut - 20))
    mine = rng.integers(0, 2**32, size=(60, d.dim), dtype=np.uint32)
    ctx.upload_table(t, mine, row0=cut - 20)
    assert np.array_equal(ctx.download_table(t, cut - 20, 60), mine)
    other = next(u for u in range(m.n_tables) if bot[u] == bot[t] and u != t)   # a neighbour in the same bank rows is untouched
    du = tabs[other]
    n_o = min(du.rows, cut)
    assert np.array_equal(ctx.download_table(other, 0, n_o), O.content_rows(O.FILL_TAGGED, 0, du.source * 1024 + du.mem_class * 256 + du.table_id, n_o, du.dim))
    idx2 = idx.copy()
    idx2[:, bot[t]] = cut - 20 + np.arange(B) % 20   # the uploaded rows inside the interleaved region come back through the gather
    rec = wk.gather_records(idx2, dense).reshape(B, m.record_len)
    seg = next(s for s in m.segments() if s.kind == fr.SEG_TABLE and s.src == t)
    assert np.array_equal(rec[:, seg.rec_offset:seg.rec_offset + d.dim], mine[np.arange(B) % 20])
    bad = idx.copy()
    bad[7, bot[t]] = cut   # valid for table t itself, not for the smallest table of its bank
    d_idx = fr.DeviceBuffer.from_numpy(ctx, bad)
    d_dense = fr.DeviceBuffer.from_numpy(ctx, dense) if dense is not None else None
    d_rec = fr.DeviceBuffer(ctx, B * m.record_len * 4)
    wk.gather_only(B, d_idx, d_dense, d_rec)
    with pytest.raises(fr.FleetRecError) as e:
        wk.sync()
    assert e.value.status == fr.FR_ERR_INDEX_RANGE
    wk.close()
    ctx.close()
