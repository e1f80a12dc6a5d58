"""ctypes front-end of the CPU oracle (oracle/fleetrec_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT: importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

The model structure comes straight from tests/golden/registry_{47,98,377}.json (extracted from the
reference's KERNEL text by oracle/tools/extract_registry.py).  The product's csrc/registry_data.inc is
generated from the SAME JSON (tools/gen_registry.py), so oracle-vs-product agreement checks the two
formulations (bank memories here, word descriptors there), NOT the extraction.  What checks the
extraction: SURVEY's prose rules (tests/test_registry.py::test_wire_order_matches_survey_rules) and, for
the table -> bank map, a second witness taken from the FPGA host programs and the linker's sp= map
(oracle/tools/extract_host_witness.py -> tests/golden/host_witness_*.json,
tests/test_registry.py::test_host_side_witness_agrees_with_the_kernel_walk).
"""
import ctypes
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
LIB = os.path.join(HERE, "liboracle.so")

CLS = {"HBM": 0, "DDR": 1, "PLRAM": 2}
FILL_MEMORY, FILL_EVEN_ODD, FILL_HASH, FILL_TAGGED = -1, 0, 1, 2


def build(force=False):
    src = os.path.join(HERE, "fleetrec_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_content_bits.restype = ctypes.c_uint32
        _lib.oracle_content_bits.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32]
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def uid_of(source, cls, tid):
    return source * 1024 + CLS[cls] * 256 + tid


class Half:
    """One FPGA kernel (embedding_N_krnl) as the reference wires it."""

    def __init__(self, n, source=0):
        self.reg = json.load(open(os.path.join(GOLD, "registry_%d.json" % n)))
        self.source = source
        banks = self.reg["banks"]
        self.n_banks = len(banks)
        self.bank_ntab = np.array([len(b["tables"]) for b in banks], dtype=np.int32)
        rounds = [(bi, t) for bi, b in enumerate(banks) for t in b["tables"]]
        self.rounds = rounds
        self.tab_addr = np.array([t["addr_axi"] for _, t in rounds], dtype=np.int64)
        self.tab_axi = np.array([t["axi_words"] for _, t in rounds], dtype=np.int32)
        self.tab_uid = np.array([uid_of(source, t["class"], t["id"]) for _, t in rounds], dtype=np.uint32)
        self.tab_rows = np.array([t["rows"] for _, t in rounds], dtype=np.int64)
        self.rec_bank = np.array([r[0] for r in self.reg["record"]], dtype=np.int32)
        self.rec_k = np.array([r[1] for r in self.reg["record"]], dtype=np.int32)
        self.n_rec_words = len(self.rec_bank)
        self.record_len = 4 * self.n_rec_words
        # wire order of tables: order of first appearance in the record -> index into `rounds`
        first = np.concatenate([[0], np.cumsum(self.bank_ntab)])
        order, seen = [], set()
        for bi, k in self.reg["record"]:
            acc = 0
            for r in range(first[bi], first[bi + 1]):
                if k < acc + self.tab_axi[r]:
                    if r not in seen:
                        seen.add(r)
                        order.append(r)
                    break
                acc += self.tab_axi[r]
        assert len(order) == len(rounds)
        self.wire_to_round = np.array(order, dtype=np.int64)  # wire position -> flattened (bank, round)
        self.idx_random = np.array(self.reg["idx_random"], dtype=np.int32)

    @property
    def n_tables(self):
        return len(self.rounds)

    def banks_in_wire_order(self):
        """Bank numbers (positions in the registry's bank list) in order of first appearance in the record -- the column order of a
        per-bank index row as a host that lists tables in wire order would number them."""
        order = []
        for r in self.wire_to_round:
            b = self.rounds[r][0]
            if b not in order:
                order.append(b)
        assert len(order) == self.n_banks
        return order

    def bank_min_rows(self):
        """Per bank (registry order): the smallest TABLE_SIZE among its tables = the range over which ONE index is valid for every
        round of the bank (beyond it the reference reads the next table's rows, embedding_47_krnl.cpp:927-933)."""
        first = np.concatenate([[0], np.cumsum(self.bank_ntab)])
        return np.array([self.tab_rows[first[b]:first[b + 1]].min() for b in range(self.n_banks)], dtype=np.int64)

    def bank_images(self, fill_rows_fn):
        """Materialise the card's bank memories like host.cpp does: table t's row r sits at
        bank word ADDR_AXI + r*AXI_PADDED_SIZE.  fill_rows_fn(round_index, table_json, rows) ->
        uint32 [rows][dim].  Only for small (row-capped) cases."""
        imgs = []
        first = np.concatenate([[0], np.cumsum(self.bank_ntab)])
        for bi, b in enumerate(self.reg["banks"]):
            need = max(t["addr_axi"] + t["rows"] * t["axi_words"] for t in b["tables"])
            img = np.zeros((need, 4), dtype=np.uint32)
            for r, t in zip(range(first[bi], first[bi + 1]), b["tables"]):
                rows = fill_rows_fn(r, t, t["rows"])
                img[t["addr_axi"]:t["addr_axi"] + t["rows"] * t["axi_words"]] = rows.reshape(-1, 4)
            imgs.append(img)
        return imgs

    def bank_images_native(self, content_mode, seed=0):
        """Full-size bank memory images in host RAM, filled by the C side (OpenMP) -- what host.cpp migrates to the card
        (host.cpp:324-423,739-749).  -> list of uint8 arrays, one per bank.  Model-A: 1.4 GB, Model-B: 15 GB."""
        imgs = []
        first = np.concatenate([[0], np.cumsum(self.bank_ntab)])
        for bi, b in enumerate(self.reg["banks"]):
            need = max(t["addr_axi"] + t["rows"] * t["axi_words"] for t in b["tables"])
            img = np.zeros(need * 16, dtype=np.uint8)
            for r, t in zip(range(first[bi], first[bi + 1]), b["tables"]):
                lib().oracle_fill_bank_table(ctypes.c_int(content_mode), ctypes.c_uint32(seed), ctypes.c_uint32(int(self.tab_uid[r])),
                                             ctypes.c_int64(t["addr_axi"]), ctypes.c_int(t["axi_words"]), ctypes.c_int64(t["rows"]),
                                             _p(img, ctypes.c_uint8))
            imgs.append(img)
        return imgs

    def gather_direct(self, idx, per_round, bank_images, out=None):
        """The memory-resident CPU-baseline gather (oracle_gather_banks_direct): same arguments and result as
        gather(..., content_mode=FILL_MEMORY, bank_images=...)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        B = idx.shape[0]
        assert idx.shape[1] == (len(self.rounds) if per_round else self.n_banks)
        if out is None:
            out = np.empty((B, self.record_len), dtype=np.uint32)
        ptrs = (ctypes.c_void_p * self.n_banks)(*[im.ctypes.data for im in bank_images])
        lib().oracle_gather_banks_direct(
            ctypes.c_int(self.n_banks), _p(self.bank_ntab, ctypes.c_int32), _p(self.tab_addr, ctypes.c_int64),
            _p(self.tab_axi, ctypes.c_int32), ptrs, ctypes.c_int(self.n_rec_words), _p(self.rec_bank, ctypes.c_int32),
            _p(self.rec_k, ctypes.c_int32), _p(idx, ctypes.c_int32), ctypes.c_int(1 if per_round else 0),
            ctypes.c_int64(B), _p(out, ctypes.c_uint8))
        return out

    def gather(self, idx, per_round, content_mode, seed=0, bank_images=None):
        """idx: int32 [B][n_banks] (per_round=False) or [B][n_rounds] in flattened (bank, round) order.
        -> uint32 [B][record_len]"""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        B = idx.shape[0]
        assert idx.shape[1] == (len(self.rounds) if per_round else self.n_banks)
        out = np.empty((B, self.record_len), dtype=np.uint32)
        if bank_images is not None:
            ptrs = (ctypes.c_void_p * self.n_banks)(*[im.ctypes.data for im in bank_images])
        else:
            ptrs = (ctypes.c_void_p * self.n_banks)()
        lib().oracle_gather_banks(
            ctypes.c_int(self.n_banks), _p(self.bank_ntab, ctypes.c_int32), _p(self.tab_addr, ctypes.c_int64),
            _p(self.tab_axi, ctypes.c_int32), _p(self.tab_uid, ctypes.c_uint32), ptrs, ctypes.c_int(content_mode),
            ctypes.c_uint32(seed), ctypes.c_int(self.n_rec_words), _p(self.rec_bank, ctypes.c_int32),
            _p(self.rec_k, ctypes.c_int32), _p(idx, ctypes.c_int32), ctypes.c_int(1 if per_round else 0),
            ctypes.c_int64(B), _p(out, ctypes.c_uint8))
        return out


class OracleModel:
    """Model A / B / C as the reference composes them (C = CPU dense 64 + FPGA0 + FPGA1,
    GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/constant.h:25-27)."""

    def __init__(self, which):
        gpu = json.load(open(os.path.join(GOLD, "registry_gpu.json")))
        self.which = which
        if which == "A":
            self.halves, self.dense_len = [Half(47, 0)], 0
            self.fc = self.halves[0].reg["fc"]
        elif which == "B":
            self.halves, self.dense_len = [Half(98, 0)], 0
            self.fc = gpu["one_node"]["fc"]
        elif which == "C":
            self.halves, self.dense_len = [Half(377, 0), Half(377, 1)], gpu["three_nodes"]["len_cpu_sender"]
            self.fc = gpu["three_nodes"]["fc"]
        else:
            raise ValueError(which)
        self.record_len = self.dense_len + sum(h.record_len for h in self.halves)
        assert self.record_len == self.fc[0]
        self.n_tables = sum(h.n_tables for h in self.halves)
        # rows of every table in the product's (wire) order
        self.rows_wire = np.concatenate([h.tab_rows[h.wire_to_round] for h in self.halves])

    def src_lens(self):
        """per-source record lengths in the 3-node block order [CPU, FPGA0, FPGA1]"""
        return ([self.dense_len] if self.dense_len else []) + [h.record_len for h in self.halves]

    @property
    def n_banks(self):
        return sum(h.n_banks for h in self.halves)

    def bank_rows_wire(self):
        """Valid index range of every bank, banks in wire order of first appearance (half 0 first)."""
        return np.concatenate([h.bank_min_rows()[h.banks_in_wire_order()] for h in self.halves])

    def gather(self, idx, dense=None, content_mode=FILL_HASH, seed=0, bank_images=None, rows_wire=None, per_bank=False):
        """idx: int32 [B][n_tables] with columns in WIRE order (the product's table order), or int32 [B]
        (reference behaviour: one index per item for everything), or -- per_bank=True -- int32 [B][n_banks] with ONE index per
        bank per item (the kernel's contract: embedding_98_krnl.cpp:1026-1040), banks in wire order of first appearance.
        -> uint32 [B][record_len], SEMANTIC layout [dense | half0 | half1]."""
        idx = np.asarray(idx, dtype=np.int32)
        B = idx.shape[0]
        if per_bank:
            assert idx.ndim == 2 and idx.shape[1] == self.n_banks
            parts = []
            if self.dense_len:
                parts.append(np.ascontiguousarray(dense, dtype=np.float32).reshape(B, self.dense_len).view(np.uint32))
            col = 0
            for hi, h in enumerate(self.halves):
                sub = idx[:, col:col + h.n_banks]
                hidx = np.empty_like(sub)
                hidx[:, h.banks_in_wire_order()] = sub  # wire order of banks -> registry bank order
                col += h.n_banks
                parts.append(h.gather(hidx, False, content_mode, seed, None if bank_images is None else bank_images[hi]))
            return np.concatenate(parts, axis=1)
        parts = []
        if self.dense_len:
            d = np.ascontiguousarray(dense, dtype=np.float32).reshape(B, self.dense_len)
            parts.append(d.view(np.uint32))
        col = 0
        for hi, h in enumerate(self.halves):
            if idx.ndim == 1:
                hidx = np.repeat(idx[:, None], h.n_banks, axis=1)
                per_round = False
            else:
                sub = idx[:, col:col + h.n_tables]
                hidx = np.empty_like(sub)
                hidx[:, h.wire_to_round] = sub  # wire order -> flattened (bank, round) order
                per_round = True
            col += h.n_tables
            parts.append(h.gather(hidx, per_round, content_mode, seed,
                                  None if bank_images is None else bank_images[hi]))
        return np.concatenate(parts, axis=1)

    def fc_chain(self, records_f32, weights, acc64=True, dims=None):
        """records_f32: float32 [B][K] item-major (== column-major K x B).  weights: list of 4 float32 arrays,
        column-major H x K flattened (element (h,k) at [h + k*H]).  -> float32 [B] (OUT == 1)."""
        dims = np.array(dims if dims is not None else self.fc, dtype=np.int32)
        X = np.ascontiguousarray(records_f32, dtype=np.float32)
        B = X.shape[0]
        assert X.shape[1] == dims[0]
        for li, w in enumerate(weights):
            assert w.dtype == np.float32 and w.size == dims[li] * dims[li + 1], (li, w.size)
        scratch = np.empty(int(dims[1] + dims[2] + dims[3]) * B, dtype=np.float32)
        out = np.empty(int(dims[4]) * B, dtype=np.float32)
        f = ctypes.POINTER(ctypes.c_float)
        lib().oracle_fc_chain(_p(dims, ctypes.c_int32), ctypes.c_int64(B), _p(X, ctypes.c_float),
                              *[_p(np.ascontiguousarray(w), ctypes.c_float) for w in weights],
                              _p(scratch, ctypes.c_float), _p(out, ctypes.c_float), ctypes.c_int(1 if acc64 else 0))
        return out

    def block_records(self, records_u32):
        """SEMANTIC [B][K] -> the 3-node server's literal blocked buffer (flat uint32)."""
        lens = np.array(self.src_lens(), dtype=np.int32)
        rec = np.ascontiguousarray(records_u32, dtype=np.uint32)
        out = np.empty(rec.size, dtype=np.uint32)
        lib().oracle_block_records(ctypes.c_int(len(lens)), _p(lens, ctypes.c_int32), ctypes.c_int64(rec.shape[0]),
                                   _p(rec, ctypes.c_float), _p(out, ctypes.c_float))
        return out


def content_rows(mode, seed, uid, rows, dim, row0=0):
    """numpy restatement of content_bits for whole rows -> uint32 [rows][dim] (vectorised; used to
    cross-check the C function and the device fill kernels)."""
    r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    c = np.arange(dim, dtype=np.uint32)[None, :]
    if mode == FILL_EVEN_ODD:
        return np.where((r & np.uint64(1)) == 0, np.uint32(0x3F800000), np.uint32(0)).astype(np.uint32) + 0 * c
    if mode == FILL_TAGGED:
        source, cls, tid = uid >> 10, (uid >> 8) & 3, uid & 255
        base = np.uint32((source << 31) | (cls << 29) | (tid << 21))
        return (base | ((r & np.uint64(0xFFFF)).astype(np.uint32) << np.uint32(5)) | (c & np.uint32(31))).astype(np.uint32)

    def fmix(h):
        h = h.astype(np.uint32)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
        h ^= h >> np.uint32(16)
        return h

    with np.errstate(over="ignore"):
        h0 = fmix(np.array([(seed ^ ((uid * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF], dtype=np.uint32))[0]
        h = fmix(np.uint32(h0) ^ (r & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        h = fmix(h ^ (r >> np.uint64(32)).astype(np.uint32) ^ (c * np.uint32(0x27D4EB2F)).astype(np.uint32))
    v = (h >> np.uint32(8)).astype(np.int32).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    return v.view(np.uint32)
