"""fleetrec_amd -- Python host binding of the MI355X-native FleetRec hot path.

A thin ctypes mirror of include/fleetrec.h (the C-ABI in libfleetrec.so).  It exists for the test
harness and bench.py; the production host is the C++ driver (the counterpart of the reference's
GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c).  There is no silent CPU fallback: if the
HIP library is missing, or a context is asked for on a device that is not a visible gfx950, calls raise
FleetRecError; Context(model, device=DEVICE_CPU) asks for the library's own CPU back-end explicitly.

Vocabulary follows the reference: tables, banks, rounds, records (the per-item concatenated
feature vector the FPGA sends), workers (thread_consume), batches.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FR_LIB") or os.path.join(_HERE, "libfleetrec.so")   # FR_LIB: A/B experiments load another build of the same ABI

FR_OK, FR_ERR_INVALID, FR_ERR_NO_DEVICE, FR_ERR_OOM, FR_ERR_HIP, FR_ERR_INDEX_RANGE, FR_ERR_STATE, FR_ERR_COMM = 0, -1, -2, -3, -4, -5, -6, -7
MODEL_A, MODEL_B, MODEL_C = 0, 1, 2
FILL_EVEN_ODD, FILL_HASH, FILL_TAGGED = 0, 1, 2
WEIGHTS_ONES, WEIGHTS_UNIFORM = 0, 1
FC_FP32, FC_BF16, FC_FP8 = 0, 1, 2
LAYOUT_SEMANTIC, LAYOUT_BLOCKED = 0, 1
INDEX_PER_TABLE, INDEX_PER_ITEM, INDEX_PER_BANK = 0, 1, 2
SEG_TABLE, SEG_COPY, SEG_DENSE = 0, 1, 2
GATHER_WORD_MAJOR, GATHER_ITEM_TILE, GATHER_ITEM_TILE_DEDUP, GATHER_ITEM_TILE_DEDUP_COUNT, GATHER_WORD_MAJOR_ONE_CHUNK = 0, 1, 2, 3, 4
ABI_VERSION = 6   # include/fleetrec.h FR_ABI_VERSION this binding was written against
MEM_CLASS_NAMES = {0: "HBM", 1: "DDR", 2: "PLRAM"}


class TableDesc(ctypes.Structure):
    _fields_ = [("mem_class", ctypes.c_int32), ("table_id", ctypes.c_int32), ("source", ctypes.c_int32),
                ("dim", ctypes.c_int32), ("rows", ctypes.c_int64), ("bank", ctypes.c_int32),
                ("round", ctypes.c_int32), ("addr_axi", ctypes.c_int64)]


class Segment(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("src", ctypes.c_int32), ("src_col", ctypes.c_int32),
                ("rec_offset", ctypes.c_int32), ("len", ctypes.c_int32), ("source", ctypes.c_int32)]


class ModelDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("n_tables", ctypes.c_int32), ("n_segments", ctypes.c_int32),
                ("tables", ctypes.POINTER(TableDesc)), ("segments", ctypes.POINTER(Segment)),
                ("record_len", ctypes.c_int32), ("dense_len", ctypes.c_int32), ("fc", ctypes.c_int32 * 5),
                ("layout", ctypes.c_int32), ("index_mode", ctypes.c_int32)]


class FleetRecError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("fleetrec status %d: %s" % (status, message))
        self.status = status


_lib = None

# every symbol include/fleetrec.h, fleetrec_serving.h and fleetrec_diag.h declare (the not-gpu test checks the .so exports all of them)
ABI_SYMBOLS = [
    "fr_abi_version", "fr_last_error", "fr_device_count", "fr_cpu_set_threads", "fr_ctx_set_chain_width", "fr_ctx_chain_width", "fr_model_builtin", "fr_model_clone_scaled", "fr_model_free",
    "fr_model_table_bytes", "fr_model_index_cols", "fr_model_bank_map", "fr_ctx_set_gather_variant", "fr_ctx_gather_variant",
    "fr_ctx_gather_merged_lookups", "fr_ctx_gather_groups", "fr_comm_unique_id", "fr_comm_init_rank", "fr_comm_init_all", "fr_comm_destroy", "fr_comm_set_wait_ms",
    "fr_worker_submit_sharded", "fr_worker_calibrate_fp8_sharded", "fr_ctx_create", "fr_ctx_create_sharded", "fr_ctx_destroy", "fr_ctx_model",
    "fr_ctx_fill_tables", "fr_ctx_upload_table", "fr_ctx_download_table", "fr_ctx_set_weights", "fr_ctx_fill_weights",
    "fr_ctx_get_weights", "fr_ctx_set_fc_precision", "fr_ctx_get_fp8_exponents", "fr_ctx_set_fp8_act_exponents",
    "fr_worker_calibrate_fp8", "fr_worker_calibrate_fp8_slices", "fr_worker_push_host", "fr_worker_stage_acquire", "fr_worker_push_staged", "fr_worker_flush", "fr_worker_host_poll", "fr_worker_host_pending", "fr_ctx_set_small_block", "fr_worker_stream", "fr_worker_gather_slices", "fr_worker_fc_from_slices_lp", "fr_worker_create", "fr_worker_destroy", "fr_worker_idx_ptr",
    "fr_worker_dense_ptr", "fr_worker_score_ptr", "fr_worker_submit", "fr_worker_submit_device", "fr_worker_push_device", "fr_worker_push_device_list", "fr_worker_sync",
    "fr_worker_gather_only", "fr_worker_fc_only", "fr_worker_fc_layer_only", "fr_worker_fc_layer_repeat", "fr_worker_records_dptr", "fr_worker_features_dptr", "fr_worker_timer_start",
    "fr_worker_timer_stop_ms", "fr_device_malloc", "fr_device_free", "fr_memcpy_h2d", "fr_memcpy_d2h",
    "fr_device_synchronize", "fr_ctx_shard_info", "fr_driver_create", "fr_driver_destroy", "fr_driver_run_resident",
    "fr_driver_worker", "fr_driver_score_ring", "fr_driver_run_host", "fr_driver_run_host_streaming", "fr_driver_host_score_ring", "fr_ctx_stream_group", "fr_ctx_set_stream_group", "fr_model_shard_plan", "fr_worker_fc_from_slices", "fr_worker_last_kernel", "fr_worker_inject_fc_failure", "fr_ctx_set_lp_bank_image", "fr_ctx_lp_bank_image_bytes",
]


def lib():
    """Load libfleetrec.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FleetRecError(FR_ERR_STATE, "HIP library %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C gpu-fpga-recommendation-system_amd/csrc`" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_size_t
    pf, pi = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    sig = {
        "fr_abi_version": (i32, []), "fr_last_error": (ctypes.c_char_p, []), "fr_device_count": (i32, []), "fr_cpu_set_threads": (i32, [i32]), "fr_ctx_set_chain_width": (i32, [vp, i32]), "fr_ctx_chain_width": (i32, [vp]),
        "fr_model_builtin": (ctypes.POINTER(ModelDesc), [i32]),
        "fr_model_clone_scaled": (i32, [ctypes.POINTER(ModelDesc), ctypes.c_double, i64, i64, ctypes.POINTER(ctypes.POINTER(ModelDesc))]),
        "fr_model_free": (None, [ctypes.POINTER(ModelDesc)]),
        "fr_model_table_bytes": (i64, [ctypes.POINTER(ModelDesc)]),
        "fr_model_index_cols": (i32, [ctypes.POINTER(ModelDesc)]),
        "fr_model_bank_map": (i32, [ctypes.POINTER(ModelDesc), pi, ctypes.POINTER(ctypes.c_int64)]),
        "fr_ctx_create": (i32, [ctypes.POINTER(ModelDesc), i32, ctypes.POINTER(vp)]),
        "fr_ctx_create_sharded": (i32, [ctypes.POINTER(ModelDesc), i32, i32, i32, ctypes.POINTER(vp)]),
        "fr_ctx_destroy": (None, [vp]), "fr_ctx_model": (ctypes.POINTER(ModelDesc), [vp]),
        "fr_ctx_fill_tables": (i32, [vp, i32, u32]),
        "fr_ctx_upload_table": (i32, [vp, i32, i64, i64, vp]), "fr_ctx_download_table": (i32, [vp, i32, i64, i64, vp]),
        "fr_ctx_set_weights": (i32, [vp, i32, vp, sz]), "fr_ctx_fill_weights": (i32, [vp, i32, u32]),
        "fr_ctx_get_weights": (i32, [vp, i32, vp, sz]), "fr_ctx_set_fc_precision": (i32, [vp, i32]),
        "fr_ctx_get_fp8_exponents": (i32, [vp, vp, vp]), "fr_ctx_set_fp8_act_exponents": (i32, [vp, vp]),
        "fr_worker_push_host": (i32, [vp, i32, vp, vp, vp]), "fr_worker_stream": (vp, [vp]),
        "fr_worker_stage_acquire": (i32, [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp)]), "fr_worker_push_staged": (i32, [vp, i32, vp]),
        "fr_ctx_set_small_block": (i32, [vp, i32]), "fr_worker_flush": (i32, [vp]), "fr_worker_host_poll": (i32, [vp, ctypes.POINTER(ctypes.c_longlong)]),
        "fr_worker_host_pending": (i32, [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]),
        "fr_worker_gather_slices": (i32, [vp, i32, vp, vp, vp, i32]), "fr_worker_fc_from_slices_lp": (i32, [vp, i32, i32, i32, vp, i32, vp]),
        "fr_worker_calibrate_fp8": (i32, [vp, i32]), "fr_worker_calibrate_fp8_slices": (i32, [vp, i32, i32, i32, vp]),
        "fr_worker_create": (i32, [vp, i32, ctypes.POINTER(vp)]), "fr_worker_destroy": (None, [vp]),
        "fr_worker_idx_ptr": (pi, [vp]), "fr_worker_dense_ptr": (pf, [vp]), "fr_worker_score_ptr": (pf, [vp]),
        "fr_worker_submit": (i32, [vp, i32]), "fr_worker_submit_device": (i32, [vp, i32, vp, vp, vp]), "fr_worker_push_device": (i32, [vp, i32, vp, vp, vp]), "fr_worker_push_device_list": (i32, [vp, i32, vp, vp, vp, vp]),
        "fr_worker_sync": (i32, [vp]), "fr_worker_gather_only": (i32, [vp, i32, vp, vp, vp]),
        "fr_worker_fc_only": (i32, [vp, i32, vp, vp]), "fr_worker_fc_layer_only": (i32, [vp, i32, i32]), "fr_worker_fc_layer_repeat": (i32, [vp, i32, i32, i32]), "fr_worker_records_dptr": (vp, [vp]),
        "fr_worker_features_dptr": (vp, [vp, ctypes.POINTER(ctypes.c_int)]),
        "fr_worker_timer_start": (i32, [vp]), "fr_worker_timer_stop_ms": (i32, [vp, pf]),
        "fr_device_malloc": (i32, [vp, sz, ctypes.POINTER(vp)]), "fr_device_free": (i32, [vp, vp]),
        "fr_memcpy_h2d": (i32, [vp, vp, vp, sz]), "fr_memcpy_d2h": (i32, [vp, vp, vp, sz]),
        "fr_device_synchronize": (i32, [vp]),
        "fr_ctx_shard_info": (i32, [vp] + [ctypes.POINTER(ctypes.c_int)] * 5),
        "fr_driver_create": (i32, [vp, i32, i32, i32, ctypes.POINTER(vp)]), "fr_driver_destroy": (None, [vp]),
        "fr_driver_run_resident": (i32, [vp, i32, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), i32, ctypes.POINTER(ctypes.c_double)]),
        "fr_driver_worker": (vp, [vp, i32, i32]), "fr_driver_score_ring": (vp, [vp, i32, i32, ctypes.POINTER(ctypes.c_int)]),
        "fr_ctx_stream_group": (i32, [vp]), "fr_ctx_set_stream_group": (i32, [vp, i32]),
        "fr_ctx_set_gather_variant": (i32, [vp, i32]), "fr_ctx_gather_variant": (i32, [vp]),
        "fr_comm_unique_id": (i32, [vp]), "fr_comm_init_rank": (i32, [vp, vp, ctypes.POINTER(vp)]),
        "fr_comm_init_all": (i32, [ctypes.POINTER(vp), i32, ctypes.POINTER(vp)]), "fr_comm_destroy": (None, [vp]), "fr_comm_set_wait_ms": (i32, [vp, i32]),
        "fr_worker_submit_sharded": (i32, [vp, vp, i32]), "fr_worker_calibrate_fp8_sharded": (i32, [vp, vp, i32]),
        "fr_ctx_gather_merged_lookups": (i32, [vp, ctypes.POINTER(ctypes.c_uint64), i32]),
        "fr_ctx_gather_groups": (i32, [vp, ctypes.POINTER(ctypes.c_int)]),
        "fr_driver_run_host": (i32, [vp, i32, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), i32, ctypes.POINTER(ctypes.c_double)]),
        "fr_driver_run_host_streaming": (i32, [vp, i32, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), i32, ctypes.POINTER(ctypes.c_double)]),
        "fr_driver_host_score_ring": (vp, [vp, i32, i32, ctypes.POINTER(ctypes.c_int)]),
        "fr_model_shard_plan": (i32, [ctypes.POINTER(ModelDesc), i32, pi, pi, ctypes.POINTER(ctypes.c_int)]),
        "fr_worker_fc_from_slices": (i32, [vp, i32, i32, i32, vp, vp]),
        "fr_worker_last_kernel": (ctypes.c_char_p, [vp]), "fr_worker_inject_fc_failure": (i32, [vp, i32]),
        "fr_ctx_set_lp_bank_image": (i32, [vp, i32]), "fr_ctx_lp_bank_image_bytes": (ctypes.c_size_t, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    got = L.fr_abi_version()
    if got != ABI_VERSION:   # an FR_LIB build of another ABI would be called with the wrong contract
        raise FleetRecError(FR_ERR_STATE, "%s reports ABI version %d, this binding needs %d" % (LIB_PATH, got, ABI_VERSION))
    _lib = L
    return L


def _check(status):
    if status != FR_OK:
        raise FleetRecError(status, lib().fr_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().fr_device_count()


DEVICE_CPU = -1   # Context(model, device=DEVICE_CPU): the library's CPU back-end (fp32 only; include/fleetrec.h fr_ctx_create)


def cpu_set_threads(n=0):
    """Host threads of the CPU back-end (0 = every usable core) -> the number in use."""
    r = lib().fr_cpu_set_threads(n)
    if r < 0:
        _check(r)
    return r


class Model:
    """A model description (built-in A/B/C, or a row-scaled clone)."""

    def __init__(self, ptr, owned=False, keepalive=None):
        self._ptr, self._owned, self._keep = ptr, owned, keepalive

    @classmethod
    def builtin(cls, which):
        p = lib().fr_model_builtin(which)
        if not p:
            _check(FR_ERR_INVALID)
        return cls(p)

    @classmethod
    def from_spec(cls, spec):
        """Build a user-defined model from a plain dict (the run-time counterpart of the reference's generated
        constants.hpp, FPGA/kernel/user_krnl/embedding_47_krnl/src/hls/constants.hpp:28,505):

            {"name": "my_model",
             "tables": [{"dim": 8, "rows": 100000, "class": "HBM", "bank": 0}, ...],   # listed in record (wire) order; tables with
                                                                                # equal (class, bank) share a memory bank (default: own bank)
             "dense_len": 0,                                                    # floats per item supplied by the request
             "dense_at": 0,                                                     # table position the dense block precedes
             "pad": [{"after_table": 3, "copy_of": 1, "col": 0}],               # optional 4-float COPY pads
             "fc": [1024, 512, 256]}                                            # hidden widths; K = record length, OUT = 1

        The description is validated by the library (fr_model_clone_scaled -> fr_model_validate)."""
        tabs_in = spec["tables"]
        n = len(tabs_in)
        tabs = (TableDesc * n)()
        cls_id = {"HBM": 0, "DDR": 1, "PLRAM": 2}
        for t, d in enumerate(tabs_in):
            tabs[t] = TableDesc(mem_class=cls_id.get(d.get("class", "HBM"), 0), table_id=t % 256, source=0, dim=int(d["dim"]),
                                rows=int(d["rows"]), bank=int(d.get("bank", t)), round=0, addr_axi=0)
        dense_len, dense_at = int(spec.get("dense_len", 0)), int(spec.get("dense_at", 0))
        pads = {int(p_["after_table"]): p_ for p_ in spec.get("pad", [])}
        segs, pos, seen_dense = [], 0, False
        for t in range(n + 1):
            if dense_len and t == dense_at:
                segs.append((SEG_DENSE, -1, 0, pos, dense_len, 2))
                pos += dense_len
                seen_dense = True
            if t == n:
                break
            src_id = 1 if seen_dense else 0
            segs.append((SEG_TABLE, t, 0, pos, int(tabs_in[t]["dim"]), src_id))
            pos += int(tabs_in[t]["dim"])
            if t in pads:
                segs.append((SEG_COPY, int(pads[t]["copy_of"]), int(pads[t].get("col", 0)), pos, 4, src_id))
                pos += 4
        S = (Segment * len(segs))(*[Segment(kind=k, src=s_, src_col=c, rec_offset=o, len=l, source=sr) for k, s_, c, o, l, sr in segs])
        d = ModelDesc()
        d.name = spec.get("name", "custom").encode()[:31]
        d.n_tables, d.n_segments = n, len(segs)
        d.tables = ctypes.cast(tabs, ctypes.POINTER(TableDesc))
        d.segments = ctypes.cast(S, ctypes.POINTER(Segment))
        d.record_len, d.dense_len = pos, dense_len
        fcw = [pos] + [int(v) for v in spec["fc"]] + [1]
        if len(fcw) != 5:
            raise FleetRecError(FR_ERR_INVALID, "spec['fc'] must list exactly three hidden widths")
        for i, v in enumerate(fcw):
            d.fc[i] = v
        tmp = cls(ctypes.pointer(d), keepalive=(tabs, S, d))
        return tmp.clone()  # validated deep copy owned by the library

    def placement_report(self, l2_bytes=4 << 20, mall_bytes=256 << 20):
        """Where each table will live on MI355X when accessed uniformly, and what the gather costs per item: the
        MicroRec/FleetRec placement question (on-chip PLRAM vs HBM vs DDR banks) mapped onto L2 / Infinity Cache / HBM.
        A row costs one 128-byte line beyond L2 whatever its size (profiles/archive/r01_experiments.md)."""
        tabs = self.tables()
        order = sorted(range(len(tabs)), key=lambda t: tabs[t].rows * tabs[t].dim * 4)
        cum, out = 0, []
        for t in order:
            b = tabs[t].rows * tabs[t].dim * 4
            cum += b
            level = "L2" if cum <= 8 * l2_bytes else ("InfinityCache" if cum <= mall_bytes else "HBM")
            out.append({"table": t, "class": MEM_CLASS_NAMES[tabs[t].mem_class], "id": tabs[t].table_id, "dim": tabs[t].dim,
                        "rows": tabs[t].rows, "bytes": b, "level": level})
        rows_per_item = len(tabs)
        useful = sum(t.dim * 4 for t in tabs)
        return {"tables": sorted(out, key=lambda e: e["table"]), "table_bytes": cum, "rows_per_item": rows_per_item,
                "useful_row_bytes_per_item": useful, "line_bytes_per_item_beyond_l2": 128 * sum(1 for e in out if e["level"] != "L2"),
                "levels": {lv: sum(1 for e in out if e["level"] == lv) for lv in ("L2", "InfinityCache", "HBM")}}

    def clone(self, row_scale=1.0, min_rows=1, max_rows=0, layout=None, index_mode=None):
        out = ctypes.POINTER(ModelDesc)()
        _check(lib().fr_model_clone_scaled(self._ptr, row_scale, min_rows, max_rows, ctypes.byref(out)))
        m = Model(out, owned=True)
        if layout is not None:
            out.contents.layout = layout
        if index_mode is not None:
            out.contents.index_mode = index_mode
        return m

    def __del__(self):
        if getattr(self, "_owned", False) and self._ptr:
            lib().fr_model_free(self._ptr)
            self._ptr = None

    @property
    def desc(self):
        return self._ptr.contents

    @property
    def name(self):
        return self.desc.name.decode()

    @property
    def n_tables(self):
        return self.desc.n_tables

    @property
    def record_len(self):
        return self.desc.record_len

    @property
    def dense_len(self):
        return self.desc.dense_len

    @property
    def fc(self):
        return list(self.desc.fc)

    @property
    def idx_cols(self):
        """int32 columns of one item's index row: n_tables (PER_TABLE), 1 (PER_ITEM) or the number of banks (PER_BANK)."""
        n = lib().fr_model_index_cols(self._ptr)
        if n <= 0:
            _check(n if n < 0 else FR_ERR_INVALID)
        return n

    def bank_map(self):
        """-> (bank_of_table int32 [n_tables], bank_rows int64 [n_banks]): the memory bank (= index column in PER_BANK mode) of
        every table, banks numbered by first appearance in the table list, and the valid index range of each bank (the smallest
        row count among its tables)."""
        bot = np.empty(self.n_tables, dtype=np.int32)
        rows = np.empty(self.n_tables, dtype=np.int64)
        nb = lib().fr_model_bank_map(self._ptr, bot.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        if nb < 0:
            _check(nb)
        return bot, rows[:nb].copy()

    def index_ranges(self):
        """Exclusive upper bound of every index column in the model's index mode (uniform-index generators use this)."""
        mode = self.desc.index_mode
        if mode == INDEX_PER_TABLE:
            return self.rows()
        if mode == INDEX_PER_BANK:
            return self.bank_map()[1]
        return np.array([self.rows().min()], dtype=np.int64)

    def tables(self):
        d = self.desc
        return [d.tables[i] for i in range(d.n_tables)]

    def segments(self):
        d = self.desc
        return [d.segments[i] for i in range(d.n_segments)]

    def rows(self):
        return np.array([t.rows for t in self.tables()], dtype=np.int64)

    def table_bytes(self):
        return lib().fr_model_table_bytes(self._ptr)

    def shard_plan(self, n_shards):
        """-> (slice_offsets, slice_lens, padded_len): the table-ID sharding of the record over n_shards GPUs (host only)."""
        off = (ctypes.c_int32 * n_shards)()
        ln = (ctypes.c_int32 * n_shards)()
        pad = ctypes.c_int()
        _check(lib().fr_model_shard_plan(self._ptr, n_shards, off, ln, ctypes.byref(pad)))
        return list(off), list(ln), pad.value


    def shard_table_bytes(self, n_shards):
        """Table bytes every shard of the n_shards-way plan would hold (host only): a table belongs to the shard whose record slice its
        segments fall in."""
        offs, lens, _ = self.shard_plan(n_shards)
        tabs = self.tables()
        out = [0] * n_shards
        seen = set()
        for sg in self.segments():
            if sg.kind == SEG_DENSE:
                continue
            g = max(i for i in range(n_shards) if offs[i] <= sg.rec_offset)
            if (g, sg.src) not in seen:
                seen.add((g, sg.src))
                out[g] += int(tabs[sg.src].rows) * int(tabs[sg.src].dim) * 4
        return out

    def min_shards(self, hbm_bytes=288e9, reserve=0.10, candidates=(1, 2, 4, 8)):
        """BASELINE.json north_star's rule: tables shard by table-ID over the GPUs of a node ONLY when they do not fit one GPU's HBM.
        -> the smallest G of `candidates` whose largest shard fits (1 - reserve) * hbm_bytes, or None when even the last one does not
        (table-ID sharding cannot split a table)."""
        budget = (1.0 - reserve) * hbm_bytes
        for G in candidates:
            if G <= self.desc.n_segments and max(self.shard_table_bytes(G)) <= budget:
                return G
        return None


class DeviceBuffer:
    """Raw HBM allocation made through the C-ABI (no torch involved)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = ctypes.c_void_p()
        _check(lib().fr_device_malloc(ctx._h, self.nbytes, ctypes.byref(p)))
        self.ptr = p

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(ctx, arr.nbytes)
        b.upload(arr)
        return b

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(lib().fr_memcpy_h2d(self.ctx._h, self.ptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(lib().fr_memcpy_d2h(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().fr_device_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """Device + model + tables + weights; shared by all workers (one per GPU)."""

    def __init__(self, model, device=0, shard_rank=0, n_shards=1):
        self.model = model
        h = ctypes.c_void_p()
        _check(lib().fr_ctx_create_sharded(model._ptr, device, shard_rank, n_shards, ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().fr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fill_tables(self, mode, seed=0):
        _check(lib().fr_ctx_fill_tables(self._h, mode, seed))

    def upload_table(self, table, rows_f32, row0=0):
        a = np.ascontiguousarray(rows_f32)
        assert a.dtype in (np.float32, np.uint32)
        nrows = a.shape[0]
        _check(lib().fr_ctx_upload_table(self._h, table, row0, nrows, a.ctypes.data_as(ctypes.c_void_p)))

    def download_table(self, table, row0, nrows, dtype=np.uint32):
        dim = self.model.desc.tables[table].dim
        out = np.empty((nrows, dim), dtype=dtype)
        _check(lib().fr_ctx_download_table(self._h, table, row0, nrows, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_weights(self, layer, w_colmajor):
        w = np.ascontiguousarray(w_colmajor, dtype=np.float32).ravel()
        _check(lib().fr_ctx_set_weights(self._h, layer, w.ctypes.data_as(ctypes.c_void_p), w.size))

    def fill_weights(self, mode, seed=0):
        _check(lib().fr_ctx_fill_weights(self._h, mode, seed))

    def get_weights(self, layer):
        fc = self.model.fc
        w = np.empty(fc[layer] * fc[layer + 1], dtype=np.float32)
        _check(lib().fr_ctx_get_weights(self._h, layer, w.ctypes.data_as(ctypes.c_void_p), w.size))
        return w

    def set_fc_precision(self, precision):
        _check(lib().fr_ctx_set_fc_precision(self._h, precision))

    def fp8_exponents(self):
        """-> (act_exp[4] for X, R1, R2, R3; w_exp[3] for W1..W3): tensor T is stored as e4m3(saturate(T * 2**e))."""
        a, w = (ctypes.c_int * 4)(), (ctypes.c_int * 3)()
        _check(lib().fr_ctx_get_fp8_exponents(self._h, a, w))
        return list(a), list(w)

    def set_fp8_act_exponents(self, act_exp):
        _check(lib().fr_ctx_set_fp8_act_exponents(self._h, (ctypes.c_int * 4)(*[int(v) for v in act_exp])))

    def shard_info(self):
        v = [ctypes.c_int() for _ in range(5)]
        _check(lib().fr_ctx_shard_info(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(["shard_rank", "n_shards", "slice_offset", "slice_len", "slice_padded"], [x.value for x in v]))

    def stream_group(self):
        return lib().fr_ctx_stream_group(self._h)

    def set_stream_group(self, batches_per_launch):
        _check(lib().fr_ctx_set_stream_group(self._h, batches_per_launch))

    def set_lp_bank_image(self, on):
        """fleetrec_diag.h: 0 = the in-chain gather of a per-bank bf16 / fp8 context reads the fp32 rows (A/B and parity hook)."""
        _check(lib().fr_ctx_set_lp_bank_image(self._h, int(bool(on))))

    def lp_bank_image_bytes(self):
        return int(lib().fr_ctx_lp_bank_image_bytes(self._h))

    def set_chain_width(self, width):
        """Chain width W (1..4): a chain model's bf16 / fp8 GEMM layers take tiles covering 1 / W of the chip (fr_ctx_set_chain_width)."""
        _check(lib().fr_ctx_set_chain_width(self._h, width))

    def chain_width(self):
        """The context's chain width; 0 while undecided (the first low-precision GEMM-layer launch freezes it)."""
        return lib().fr_ctx_chain_width(self._h)

    def set_small_block(self, max_batches):
        """Host-fed blocks of at most max_batches batches take fr_worker_submit's stage launches instead of the fused kernel (latency)."""
        _check(lib().fr_ctx_set_small_block(self._h, max_batches))

    def set_gather_variant(self, variant):
        """GATHER_WORD_MAJOR (default) / GATHER_ITEM_TILE / GATHER_ITEM_TILE_DEDUP(_COUNT): which kernel fr_worker_gather_only runs."""
        _check(lib().fr_ctx_set_gather_variant(self._h, variant))

    def gather_groups(self):
        """Word ranges [starts[g], starts[g + 1]) the word-major gather deals to the 8 XCD groups (fr_ctx_gather_groups)."""
        st = (ctypes.c_int * 9)()
        _check(lib().fr_ctx_gather_groups(self._h, st))
        return [int(x) for x in st]

    def gather_merged_lookups(self, reset=True):
        v = ctypes.c_uint64()
        _check(lib().fr_ctx_gather_merged_lookups(self._h, ctypes.byref(v), 1 if reset else 0))
        return v.value

    def synchronize(self):
        _check(lib().fr_device_synchronize(self._h))

    def buffer(self, nbytes):
        return DeviceBuffer(self, nbytes)


class Worker:
    """One stream + staging buffers = one thread_consume() of the reference server."""

    def __init__(self, ctx, max_batch):
        self.ctx, self.max_batch = ctx, max_batch
        h = ctypes.c_void_p()
        _check(lib().fr_worker_create(ctx._h, max_batch, ctypes.byref(h)))
        self._h = h
        m = ctx.model
        self.idx = np.ctypeslib.as_array(lib().fr_worker_idx_ptr(h), shape=(max_batch, m.idx_cols))
        self.score = np.ctypeslib.as_array(lib().fr_worker_score_ptr(h), shape=(max_batch,))
        self.dense = (np.ctypeslib.as_array(lib().fr_worker_dense_ptr(h), shape=(max_batch, m.dense_len))
                      if m.dense_len else None)

    def close(self):
        if self._h:
            lib().fr_worker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, batch):
        _check(lib().fr_worker_submit(self._h, batch))

    def sync(self):
        _check(lib().fr_worker_sync(self._h))

    def calibrate_fp8(self, idx, dense=None):
        """fp8 chain: run this batch through the fp32 chain and derive the activation exponents from the observed maxima."""
        idx = np.asarray(idx, dtype=np.int32).reshape(len(idx), -1)
        B = idx.shape[0]
        self.idx[:B] = idx
        if self.dense is not None:
            self.dense[:B] = np.asarray(dense, dtype=np.float32).reshape(B, -1)
        _check(lib().fr_worker_calibrate_fp8(self._h, B))

    def infer(self, idx, dense=None):
        """Host-buffer path: idx int32 [B][idx_cols] (+ dense float32 [B][dense_len]) -> scores float32 [B]."""
        idx = np.asarray(idx, dtype=np.int32).reshape(len(idx), -1)
        B = idx.shape[0]
        self.idx[:B] = idx
        if self.dense is not None:
            self.dense[:B] = np.asarray(dense, dtype=np.float32).reshape(B, -1)
        self.submit(B)
        self.sync()
        return self.score[:B].copy()

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        return x.ptr if isinstance(x, DeviceBuffer) else ctypes.c_void_p(x)

    def infer_sharded(self, comm, idx, dense=None):
        """The sharded hot-loop body (collective): this rank's worker receives the WHOLE request batch; -> all `batch` scores."""
        idx = np.asarray(idx, dtype=np.int32).reshape(len(idx), -1)
        B = idx.shape[0]
        self.idx[:B] = idx
        if self.dense is not None:
            self.dense[:B] = np.asarray(dense, dtype=np.float32).reshape(B, -1)
        _check(lib().fr_worker_submit_sharded(self._h, comm._h, B))
        self.sync()
        return self.score[:B].copy()

    def submit_sharded(self, comm, batch):
        """fr_worker_submit_sharded on the rows already in self.idx / self.dense (asynchronous; follow with sync())."""
        _check(lib().fr_worker_submit_sharded(self._h, comm._h, int(batch)))

    def inject_fc_failure(self, steps):
        """fleetrec_diag.h test hook: this worker's next `steps` sharded steps report a failed FC chain (failure protocol, kind (2))."""
        _check(lib().fr_worker_inject_fc_failure(self._h, int(steps)))

    def calibrate_fp8_sharded(self, comm, idx, dense=None):
        idx = np.asarray(idx, dtype=np.int32).reshape(len(idx), -1)
        B = idx.shape[0]
        self.idx[:B] = idx
        if self.dense is not None:
            self.dense[:B] = np.asarray(dense, dtype=np.float32).reshape(B, -1)
        _check(lib().fr_worker_calibrate_fp8_sharded(self._h, comm._h, B))

    def submit_device(self, batch, d_idx, d_dense, d_scores):
        _check(lib().fr_worker_submit_device(self._h, batch, self._ptr(d_idx), self._ptr(d_dense), self._ptr(d_scores)))

    def push_host(self, idx, dense, scores_out):
        """Host-fed streaming push: idx int32 [B][idx_cols] (+ dense float32 [B][dense_len]); scores_out: a float32 numpy array of
        at least B elements that stays alive until sync() -- it receives the scores."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        B = idx.shape[0]
        d = None if dense is None else np.ascontiguousarray(dense, dtype=np.float32)
        assert scores_out.dtype == np.float32 and scores_out.flags["C_CONTIGUOUS"] and scores_out.size >= B
        _check(lib().fr_worker_push_host(self._h, B, idx.ctypes.data_as(ctypes.c_void_p), None if d is None else d.ctypes.data_as(ctypes.c_void_p),
                                         scores_out.ctypes.data_as(ctypes.c_void_p)))

    def stage_acquire(self, batch):
        """Zero-copy host-fed push, step 1: numpy views (idx int32 [batch][idx_cols], dense float32 [batch][dense_len] or None) of the
        worker's pinned staging slot for the NEXT pushed batch; fill them, then push_staged()."""
        pi, pd = ctypes.c_void_p(), ctypes.c_void_p()
        _check(lib().fr_worker_stage_acquire(self._h, batch, ctypes.byref(pi), ctypes.byref(pd)))
        m = self.ctx.model
        idx = np.ctypeslib.as_array(ctypes.cast(pi, ctypes.POINTER(ctypes.c_int32)), shape=(batch, m.idx_cols))
        dense = np.ctypeslib.as_array(ctypes.cast(pd, ctypes.POINTER(ctypes.c_float)), shape=(batch, m.dense_len)) if (m.dense_len and pd.value) else None
        return idx, dense

    def push_staged(self, batch, scores_out):
        """Step 2: queue the batch written into the acquired slot; scores_out as for push_host()."""
        assert scores_out.dtype == np.float32 and scores_out.flags["C_CONTIGUOUS"] and scores_out.size >= batch
        _check(lib().fr_worker_push_staged(self._h, batch, scores_out.ctypes.data_as(ctypes.c_void_p)))

    def flush(self):
        """Launch what is queued on the worker now, without waiting (fr_worker_flush)."""
        _check(lib().fr_worker_flush(self._h))

    def host_pending(self):
        """-> (queued, in_flight, blocks_in_flight): host-fed batches in the block being filled / launched and not delivered yet / launched blocks."""
        q, f, nb = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _check(lib().fr_worker_host_pending(self._h, ctypes.byref(q), ctypes.byref(f), ctypes.byref(nb)))
        return q.value, f.value, nb.value

    def host_poll(self):
        """Deliver the scores of finished host-fed blocks (no waiting); -> host-fed batches delivered so far, in push order."""
        n = ctypes.c_longlong()
        _check(lib().fr_worker_host_poll(self._h, ctypes.byref(n)))
        return n.value

    def push_device(self, batch, d_idx, d_dense, d_scores):
        _check(lib().fr_worker_push_device(self._h, batch, self._ptr(d_idx), self._ptr(d_dense), self._ptr(d_scores)))

    def make_push_list(self, batches, d_idx, d_dense, d_scores):
        """Pre-built argument arrays for push_device_list: batches [n] ints, d_idx / d_scores [n] DeviceBuffers (or pointers), d_dense [n] or None."""
        n = len(batches)
        vp = ctypes.c_void_p
        arr = lambda xs: (vp * n)(*[self._ptr(x) for x in xs])
        return (n, (ctypes.c_int * n)(*[int(b) for b in batches]), arr(d_idx), arr(d_dense) if d_dense is not None else None, arr(d_scores),
                (d_idx, d_dense, d_scores))   # (the buffers stay referenced as long as the list does)

    def push_device_list(self, plist):
        """n consecutive push_device calls in one native call (fr_worker_push_device_list); plist from make_push_list."""
        _check(lib().fr_worker_push_device_list(self._h, plist[0], plist[1], plist[2], plist[3], plist[4]))

    def gather_only(self, batch, d_idx, d_dense, d_records):
        _check(lib().fr_worker_gather_only(self._h, batch, self._ptr(d_idx), self._ptr(d_dense), self._ptr(d_records)))

    def fc_only(self, batch, d_records, d_scores):
        _check(lib().fr_worker_fc_only(self._h, batch, self._ptr(d_records), self._ptr(d_scores)))

    def fc_layer_only(self, batch, layer):
        _check(lib().fr_worker_fc_layer_only(self._h, batch, layer))

    def fc_layer_repeat(self, batch, layer, n):
        """n launches of one layer back to back from one native call (fr_worker_fc_layer_repeat)."""
        _check(lib().fr_worker_fc_layer_repeat(self._h, batch, layer, n))

    def last_kernel(self):
        """The kernel (instantiation included) of this worker's most recent fused / layer / gather launch (fr_worker_last_kernel)."""
        return (lib().fr_worker_last_kernel(self._h) or b"").decode()

    def fc_from_slices(self, batch_total, item0, n_items, d_gathered, d_scores):
        _check(lib().fr_worker_fc_from_slices(self._h, batch_total, item0, n_items, self._ptr(d_gathered), self._ptr(d_scores)))

    def stream_ptr(self):
        """The worker's hipStream_t as an integer (torch.cuda.ExternalStream(ptr) orders torch streams against it)."""
        return lib().fr_worker_stream(self._h)

    def gather_slices(self, batch, d_idx, d_dense, d_slice, transport):
        _check(lib().fr_worker_gather_slices(self._h, batch, self._ptr(d_idx), self._ptr(d_dense), self._ptr(d_slice), transport))

    def fc_from_slices_lp(self, batch_total, item0, n_items, d_gathered, transport, d_scores):
        _check(lib().fr_worker_fc_from_slices_lp(self._h, batch_total, item0, n_items, self._ptr(d_gathered), transport, self._ptr(d_scores)))

    def calibrate_fp8_slices(self, batch_total, item0, n_items, d_gathered):
        _check(lib().fr_worker_calibrate_fp8_slices(self._h, batch_total, item0, n_items, self._ptr(d_gathered)))

    def records_dptr(self):
        return lib().fr_worker_records_dptr(self._h)

    def features(self, batch, bf16=False, fp8=False):
        """Feature-major activations of the last submit(), un-packed from the device's q4 layout Xq[k/4][m][k%4]
        -> uint32 [record_len][batch] (debug/parity hook for the pipeline's own gather stage)."""
        ld_max = ctypes.c_int()
        p = lib().fr_worker_features_dptr(self._h, ctypes.byref(ld_max))
        K = self.ctx.model.record_len
        ld = (batch + 31) // 32 * 32
        if fp8:  # q16 layout Xf[k/16][m][k%16] of e4m3 bytes, K zero-padded to a multiple of 64 -> uint8 [KP][batch]
            KP = (K + 63) // 64 * 64
            out = np.empty(KP * ld, dtype=np.uint8)
            _check(lib().fr_memcpy_d2h(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p), out.nbytes))
            return out.reshape(KP // 16, ld, 16).transpose(0, 2, 1).reshape(KP, ld)[:, :batch]
        if bf16:  # q8 layout Xh[k/8][m][k%8] of bf16 -> uint16 [record_len][batch]
            out = np.empty(K * ld, dtype=np.uint16)
            _check(lib().fr_memcpy_d2h(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p), out.nbytes))
            return out.reshape(K // 8, ld, 8).transpose(0, 2, 1).reshape(K, ld)[:, :batch]
        out = np.empty(K * ld, dtype=np.uint32)
        _check(lib().fr_memcpy_d2h(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p), out.nbytes))
        return out.reshape(K // 4, ld, 4).transpose(0, 2, 1).reshape(K, ld)[:, :batch]

    def timer_start(self):
        _check(lib().fr_worker_timer_start(self._h))

    def timer_stop_ms(self):
        ms = ctypes.c_float()
        _check(lib().fr_worker_timer_stop_ms(self._h, ctypes.byref(ms)))
        return ms.value

    # convenience used by the parity tests --------------------------------------------------------
    def gather_records(self, idx, dense=None):
        """-> uint32 [flat B*K] record buffer in the model's layout (bit copy of what the kernel wrote)."""
        ctx, m = self.ctx, self.ctx.model
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int32).reshape(len(idx), -1))
        B = idx.shape[0]
        d_idx = DeviceBuffer.from_numpy(ctx, idx)
        d_dense = DeviceBuffer.from_numpy(ctx, np.asarray(dense, dtype=np.float32)) if m.dense_len else None
        info = ctx.shard_info()
        n = B * info["slice_padded"]
        d_rec = DeviceBuffer(ctx, n * 4)
        self.gather_only(B, d_idx, d_dense, d_rec)
        self.sync()
        return d_rec.download(np.uint32, n)

    def fc_scores(self, records_f32):
        ctx = self.ctx
        rec = np.ascontiguousarray(records_f32, dtype=np.float32)
        B = rec.size // ctx.model.record_len
        d_rec = DeviceBuffer.from_numpy(ctx, rec)
        d_sc = DeviceBuffer(ctx, B * 4)
        self.fc_only(B, d_rec, d_sc)
        self.sync()
        return d_sc.download(np.float32, B)


class Comm:
    """One rank's communicator over the shards of a table-sharded model (fr_comm_*): RCCL for GPU contexts, the in-process host
    exchange for CPU contexts (init_all only)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _check(lib().fr_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, ctx, unique_id):
        h = ctypes.c_void_p()
        _check(lib().fr_comm_init_rank(ctx._h, ctypes.create_string_buffer(unique_id, 128), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def init_all(cls, ctxs):
        n = len(ctxs)
        arr = (ctypes.c_void_p * n)(*[c._h for c in ctxs])
        out = (ctypes.c_void_p * n)()
        _check(lib().fr_comm_init_all(arr, n, out))
        return [cls(ctypes.c_void_p(out[i])) for i in range(n)]

    def set_wait_ms(self, ms):
        """Bound of fr_worker_sync's wait for a sharded step's collectives on this rank (default 60 s)."""
        _check(lib().fr_comm_set_wait_ms(self._h, int(ms)))

    def close(self):
        if self._h:
            lib().fr_comm_destroy(self._h)
            self._h = None


class Driver:
    """The reference server's main() + thread_consume() batch loop (without sockets), natively threaded."""

    def __init__(self, ctx, n_threads, depth, max_batch):
        self.ctx, self.n_threads, self.depth = ctx, n_threads, depth
        h = ctypes.c_void_p()
        _check(lib().fr_driver_create(ctx._h, n_threads, depth, max_batch, ctypes.byref(h)))
        self._h = h

    def run_resident(self, batch, total_batches, idx_pool, dense_pool=None):
        """idx_pool / dense_pool: lists of DeviceBuffer.  -> elapsed seconds."""
        n = len(idx_pool)
        ip = (ctypes.c_void_p * n)(*[b.ptr.value for b in idx_pool])
        dp = (ctypes.c_void_p * n)(*[b.ptr.value for b in dense_pool]) if dense_pool else None
        el = ctypes.c_double()
        _check(lib().fr_driver_run_resident(self._h, batch, total_batches, ip, dp, n, ctypes.byref(el)))
        return el.value

    def score_ring(self, thread, slot, max_batch):
        """-> float32 [ring_len][max_batch]: the scores of the last ring_len batches pushed to worker (thread, slot)."""
        n = ctypes.c_int()
        p = lib().fr_driver_score_ring(self._h, thread, slot, ctypes.byref(n))
        out = np.empty((n.value, max_batch), dtype=np.float32)
        _check(lib().fr_memcpy_d2h(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p), out.nbytes))
        return out

    def run_host(self, batch, total_batches, idx_pool_np, dense_pool_np=None, streaming=False):
        """idx_pool_np / dense_pool_np: lists of C-contiguous numpy arrays in host memory.  -> elapsed seconds (PCIe-inclusive).
        streaming=False: submit + sync per batch (the reference's sequence); True: fr_worker_push_host blocks."""
        n = len(idx_pool_np)
        keep = [np.ascontiguousarray(a, dtype=np.int32) for a in idx_pool_np]
        ip = (ctypes.c_void_p * n)(*[a.ctypes.data for a in keep])
        dp = None
        if dense_pool_np:
            keepd = [np.ascontiguousarray(a, dtype=np.float32) for a in dense_pool_np]
            dp = (ctypes.c_void_p * n)(*[a.ctypes.data for a in keepd])
        el = ctypes.c_double()
        _check((lib().fr_driver_run_host_streaming if streaming else lib().fr_driver_run_host)(self._h, batch, total_batches, ip, dp, n, ctypes.byref(el)))
        return el.value

    def host_score_ring(self, thread, slot, max_batch):
        """-> float32 [ring_len][max_batch] (host memory): scores of the last ring_len batches run_host(streaming=True) gave worker (thread, slot)."""
        n = ctypes.c_int()
        p = lib().fr_driver_host_score_ring(self._h, thread, slot, ctypes.byref(n))
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(n.value, max_batch)).copy()

    def close(self):
        if self._h:
            lib().fr_driver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
