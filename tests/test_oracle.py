"""Pins the CPU oracle (oracle/fleetrec_oracle.c) against everything the reference itself fixes for the
hot path:

  * the deterministic data pattern: even rows 1.0f / odd rows 0.0f (host.cpp:66-88) + the 32 fixed
    indices of load_access_idx => item j's record is all 0x3f800000 when idx_random[j] is even, else all 0;
  * the README known answers of the GPU server (README.md:7-11): all-ones input and weights,
    K=512 -> 68719476736, K=1024 -> 137438953472; and the same closed form for Models A/B/C;
  * structure: memory-image gather (bank words at ADDR_AXI + row*AXI_PADDED_SIZE, as the FPGA host lays
    them out) == procedural gather; one index per bank reused for every round; no bounds checks.
"""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# per-item "record is all ones" flags for the 32 fixed indices (fixture (2))
ALL_ONES_FLAGS = [0, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1, 1, 1, 1, 1]
ONE = 0x3F800000


def test_reference_pattern_known_answer(O):
    for which in "ABC":
        om = O.OracleModel(which)
        idx = om.halves[0].idx_random
        assert [int(i % 2 == 0) for i in idx] == ALL_ONES_FLAGS
        dense = np.ones((32, om.dense_len), np.float32) if om.dense_len else None
        rec = om.gather(idx, dense=dense, content_mode=O.FILL_EVEN_ODD)
        assert rec.shape == (32, om.record_len)
        emb = rec[:, om.dense_len:]
        for j in range(32):
            want = ONE if ALL_ONES_FLAGS[j] else 0
            assert (emb[j] == want).all()
        # batch_num repetitions of the same 32 items (load_access_idx loops batch_num times)
        rec2 = om.gather(np.tile(idx, 3), dense=None if dense is None else np.tile(dense, (3, 1)), content_mode=O.FILL_EVEN_ODD)
        assert np.array_equal(rec2[64:], rec)


@pytest.mark.parametrize("n", [47, 98, 377])
def test_memory_image_equals_procedural(O, n):
    """Bank images laid out like the FPGA host does (table rows at ADDR_AXI + r*AXI words) give the same
    records as the on-the-fly content function, for tagged contents and per-bank random indices."""
    h = O.Half(n)
    cap = 300
    # shrink the tables (keep ADDR_AXI: untouched pages of the np.zeros images stay unmapped)
    for b in h.reg["banks"]:
        for t in b["tables"]:
            t["rows"] = min(t["rows"], cap)
    rng = np.random.default_rng(n)

    def fill(r, t, rows):
        return O.content_rows(O.FILL_TAGGED, 0, int(h.tab_uid[r]), rows, 4 * t["axi_words"])

    imgs = h.bank_images(fill)
    B = 50
    idx_bank = rng.integers(0, 100, size=(B, h.n_banks), dtype=np.int32)  # every table has >= 100 rows
    a = h.gather(idx_bank, False, O.FILL_MEMORY, bank_images=imgs)
    b = h.gather(idx_bank, False, O.FILL_TAGGED)
    assert np.array_equal(a, b)
    # decode the tags: every float must come from the (table, row, col) the wire map says
    first = np.concatenate([[0], np.cumsum(h.bank_ntab)])
    for w, (bi, k) in enumerate(h.reg["record"]):
        acc = 0
        for r in range(first[bi], first[bi + 1]):
            if k < acc + h.tab_axi[r]:
                break
            acc += h.tab_axi[r]
        uid = int(h.tab_uid[r])
        for c in range(4):
            v = a[:, 4 * w + c]
            assert ((v >> 21) & 0xFF == (uid & 255)).all() and ((v >> 29) & 3 == ((uid >> 8) & 3)).all()
            assert ((v >> 5) & 0xFFFF == idx_bank[:, bi]).all()
            assert ((v & 31) == 4 * (k - acc) + c).all()
    # per-round (per-table) indices with all rounds of a bank equal == the per-bank form
    idx_round = idx_bank[:, np.repeat(np.arange(h.n_banks), h.bank_ntab)]
    assert np.array_equal(h.gather(idx_round, True, O.FILL_TAGGED), a)


def test_no_bounds_check_reads_next_table(O):
    """F10: idx >= rows silently reads whatever follows in the bank image (embedding_98_krnl.cpp:1026-1040)."""
    h = O.Half(98)
    for b in h.reg["banks"]:
        for t in b["tables"]:
            t["rows"] = min(t["rows"], 200)
    # make PLRAM0's second table start right after the first one
    bank = next(b for b in h.reg["banks"] if b["name"] == "PLRAM0")
    t0, t1 = bank["tables"]
    t1["addr_axi"] = t0["addr_axi"] + t0["rows"] * t0["axi_words"]
    h.tab_addr = np.array([t["addr_axi"] for _, t in h.rounds], dtype=np.int64)

    def fill(r, t, rows):
        return O.content_rows(O.FILL_TAGGED, 0, int(h.tab_uid[r]), rows, 4 * t["axi_words"])

    imgs = h.bank_images(fill)
    idx = np.full((1, h.n_banks), 5, dtype=np.int32)
    bi = [b["name"] for b in h.reg["banks"]].index("PLRAM0")
    idx[0, bi] = t0["rows"] + 3  # past the end of table 0 -> row 3 of table 1 (same AXI width)
    rec = h.gather(idx, False, O.FILL_MEMORY, bank_images=imgs)
    assert t0["axi_words"] == t1["axi_words"] == 1
    v = rec[0, 0:4]  # record floats [0,4) = PLRAM0 round 0
    assert ((v >> 21) & 0xFF == t1["id"]).all() and ((v >> 5) & 0xFFFF == 3).all()


def test_fc_known_answers(O):
    gpu = json.load(open(os.path.join(GOLD, "registry_gpu.json")))
    cases = [(ka["fc"], ka["score"]) for ka in gpu["known_answers"]]
    assert cases[0][1] == 2 ** 36 and cases[1][1] == 2 ** 37
    cases += [([352, 1024, 512, 256, 1], 352 * 2 ** 27), ([880, 1024, 512, 256, 1], 880 * 2 ** 27),
              ([3968, 2048, 512, 256, 1], 3968 * 2 ** 28)]
    assert cases[2][1] == 47244640256 and cases[3][1] == 118111600640 and cases[4][1] == 1065151889408
    om = O.OracleModel("A")
    for dims, want in cases:
        ws = [np.ones(dims[i] * dims[i + 1], np.float32) for i in range(4)]
        X = np.ones((5, dims[0]), np.float32)
        for acc64 in (True, False):
            out = om.fc_chain(X, ws, acc64=acc64, dims=dims)
            assert out.dtype == np.float32 and (out == np.float32(want)).all() and float(np.float32(want)) == want


def test_fc_matches_numpy_float64(O):
    om = O.OracleModel("A")
    rng = np.random.default_rng(0)
    dims = om.fc
    ws = [(rng.uniform(-1, 1, dims[i] * dims[i + 1]) / np.sqrt(dims[i])).astype(np.float32) for i in range(4)]
    X = rng.uniform(-1, 1, (33, dims[0])).astype(np.float32)
    r = X.astype(np.float64)
    for i in range(4):
        W = ws[i].astype(np.float64).reshape(dims[i], dims[i + 1])  # [k][h] == column-major H x K
        r = (r @ W).astype(np.float32).astype(np.float64)           # fp32 intermediates, as on the reference path
    got = om.fc_chain(X, ws, acc64=True)
    assert np.allclose(got, r[:, 0], rtol=1e-6, atol=1e-7)
    got32 = om.fc_chain(X, ws, acc64=False)
    assert np.abs(got32 - got).max() <= 1e-4 * np.abs(got).max()


def test_chained_known_answer(O):
    """records from the reference pattern -> FC with all-ones weights: score = K*H1*H2*H3 for even idx, 0 for odd."""
    for which, val in (("A", 352 * 2 ** 27), ("B", 880 * 2 ** 27), ("C", 3968 * 2 ** 28)):
        om = O.OracleModel(which)
        idx = om.halves[0].idx_random
        dense = np.tile(np.where(idx % 2 == 0, 1.0, 0.0).astype(np.float32)[:, None], (1, om.dense_len)) if om.dense_len else None
        rec = om.gather(idx, dense=dense, content_mode=O.FILL_EVEN_ODD)
        ws = [np.ones(om.fc[i] * om.fc[i + 1], np.float32) for i in range(4)]
        s = om.fc_chain(rec.view(np.float32), ws)
        assert np.array_equal(s, np.where(idx % 2 == 0, np.float32(val), np.float32(0)))


def test_blocked_layout(O):
    om = O.OracleModel("C")
    B = 7
    rec = np.arange(B * om.record_len, dtype=np.uint32).reshape(B, om.record_len)
    blk = om.block_records(rec)
    lens = om.src_lens()
    assert lens == [64, 1952, 1952]
    off, pos = 0, 0
    for L in lens:
        assert np.array_equal(blk[pos:pos + B * L].reshape(B, L), rec[:, off:off + L])
        off += L
        pos += B * L


def test_content_rows_matches_c(O):
    for mode in (O.FILL_EVEN_ODD, O.FILL_HASH, O.FILL_TAGGED):
        uid = O.uid_of(1, "PLRAM", 43)
        a = O.content_rows(mode, 0xF1EE7, uid, 257, 32, row0=2 ** 32 - 100)
        for r, c in ((0, 0), (1, 31), (99, 0), (100, 5), (101, 6), (256, 13)):
            assert int(a[r, c]) == O.lib().oracle_content_bits(mode, 0xF1EE7, uid, 2 ** 32 - 100 + r, c)
    h = O.content_rows(O.FILL_HASH, 1, 5, 4096, 32).view(np.float32)
    assert h.min() >= -1.0 and h.max() < 1.0 and abs(h.mean()) < 0.02 and 0.3 < h.var() < 0.37


def test_memory_resident_gather_matches_bank_oracle(O):
    """The cpu_baseline leg's memory-resident gather (bank images in host RAM filled like host.cpp:66-88 + one 16-byte copy per
    record word) == oracle_gather_banks on procedural contents, per-table and per-bank index forms, Model-A at full size (1.4 GB)."""
    h = O.Half(47, 0)
    imgs = h.bank_images_native(O.FILL_HASH, 0xF1EE7)
    assert sum(im.nbytes for im in imgs) == 1414678400
    rng = np.random.default_rng(3)
    B = 300
    idx = (rng.random((B, h.n_tables)) * h.tab_rows[None, :]).astype(np.int32)
    idx[0], idx[1] = 0, h.tab_rows - 1
    want = h.gather(idx, True, O.FILL_HASH, 0xF1EE7)
    assert np.array_equal(h.gather_direct(idx, True, imgs), want)
    assert np.array_equal(h.gather(idx, True, O.FILL_MEMORY, bank_images=imgs), want)
    bidx = (rng.random((B, h.n_banks)) * h.bank_min_rows()[None, :]).astype(np.int32)
    assert np.array_equal(h.gather_direct(bidx, False, imgs), h.gather(bidx, False, O.FILL_HASH, 0xF1EE7))


def test_per_bank_oracle_is_the_per_table_oracle_with_shared_indices(O):
    """One index per bank reused by every round (embedding_98_krnl.cpp:1026-1040) == the per-table form fed that index for every
    table of the bank; Model-B and one Model-C half, tagged contents."""
    for n in (98, 377):
        h = O.Half(n, 0)
        rng = np.random.default_rng(n)
        B = 64
        bidx = (rng.random((B, h.n_banks)) * h.bank_min_rows()[None, :]).astype(np.int32)
        bank_of_round = np.array([b for b, _ in h.rounds])
        assert np.array_equal(h.gather(bidx, False, O.FILL_TAGGED), h.gather(bidx[:, bank_of_round], True, O.FILL_TAGGED))


def test_golden_tagged_records_fixture(O, fr):
    """tests/golden/records_*.bin (committed bytes, made by tests/golden/make_records.py): the oracle reproduces them, and every
    float decodes to the table / row / column the product's segment list says belongs at that position of the record."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_records", os.path.join(GOLD, "make_records.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    for which, fname in mk.FILES.items():
        idx, dense, rec = mk.read(os.path.join(GOLD, fname))
        om = O.OracleModel(which)
        assert rec.shape == (32, om.record_len) and idx.shape == (32, om.n_tables)
        assert np.array_equal(om.gather(idx, dense=dense if om.dense_len else None, content_mode=O.FILL_TAGGED), rec)
        m = fr.Model.builtin({"A": fr.MODEL_A, "B": fr.MODEL_B, "C": fr.MODEL_C}[which])
        tabs = m.tables()
        for s in m.segments():
            got = rec[:, s.rec_offset:s.rec_offset + s.len]
            if s.kind == fr.SEG_DENSE:
                assert np.array_equal(got.view(np.float32), dense[:, s.src_col:s.src_col + s.len])
                continue
            t = tabs[s.src]
            tag = (np.uint32(t.source) << np.uint32(31)) | (np.uint32(t.mem_class) << np.uint32(29)) | (np.uint32(t.table_id) << np.uint32(21))
            want = tag | (idx[:, s.src].astype(np.uint32)[:, None] << np.uint32(5)) | (np.arange(s.src_col, s.src_col + s.len, dtype=np.uint32)[None, :] & np.uint32(31))
            assert np.array_equal(got, want), (which, s.src, s.rec_offset)


@pytest.mark.parametrize("which", ["A", "B", "C"])
def test_committed_fc_fixtures(O, which):
    """tests/golden/fc_cases_*.npz: random-data FC cases with committed float64 expected scores.  The generator is
    reproducible (same records, same weights, same expected numbers), and the oracle's fp64-accumulating chain -- the restatement of
    cuda_server.c:211-217,468-491 -- lands on the committed numbers."""
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fc_cases", os.path.join(ROOT, "tests", "golden", "make_fc_cases.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "fc_cases_%s.npz" % which))
    again = mk.build(which)
    assert again["records_sha256"] == str(fx["records_sha256"]) and list(again["weights_sha256"]) == list(fx["weights_sha256"])
    assert np.array_equal(again["expected"], fx["expected"])
    om = O.OracleModel(which)
    fc = [int(v) for v in fx["fc"]]
    assert fc == [int(v) for v in om.fc]
    rec = om.gather(fx["idx"], dense=fx["dense"] if om.dense_len else None, content_mode=O.FILL_HASH, seed=int(fx["seed_tables"])).view(np.float32)
    assert hashlib.sha256(rec.tobytes()).hexdigest() == str(fx["records_sha256"])
    ws = [mk.uniform_weights(int(fx["seed_weights"]), l, fc[l], fc[l + 1]) for l in range(4)]
    got = om.fc_chain(rec, ws, acc64=True)
    ref = fx["expected"]
    assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    got32 = om.fc_chain(rec, ws, acc64=False)     # fp32 accumulation in the reference's k order: inside BASELINE.json's 1e-3
    assert np.abs(got32 - ref).max() <= 1e-3 * np.abs(ref).max()
