#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, plus the evidence legs VERDICT r01 asked for.

metric  : inferences/sec at batch 256 (Model-A = embedding_47_krnl tables + FC 352-1024-512-256-1, fp32 FC), all tables
          resident in one GPU's HBM (BASELINE configs[1]); plus the embedding-gather HBM GB/s against peak on the largest
          model (Model-C, batch 4096) as the "gather" object.
step    : one pass of the hot path (index rows -> gather+pack -> 4-GEMM FC chain -> scores) over one batch of 256 synthetic
          requests whose index rows are already resident in HBM when the timed region starts (bench contract).
value   : STEADY-STATE rate: whatever --steps says, the timed region runs back-to-back batches for >= 2 s of wall clock
          (SURVEY 8(d)); `timed_batches` / `timed_s` say how many and how long, `ms_per_step` = timed_s / timed_batches.  The
          figure for EXACTLY --steps batches after --warmup batches is reported beside it as `burst` (with the driver's
          --steps 20 that is a launch-latency number: 20 batches do not even fill one fused launch group).
          `pcie_inclusive_streaming` is the same request stream starting in HOST memory with scores delivered to HOST memory
          (the reference loop's H2D / D2H included, cuda_server.c:460-461,494-495): reported next to `value`, never as it.
N > 1   : the path shards by independent request batches -> one replica per GPU, no data-path collective ("scaling":
          "weak"); value = batches all ranks processed / max-over-ranks time.  The gather half of the metric follows as
          `gather_per_bank_all_ranks` (every rank on its own Model-C replica at the same time, summed on rank 0).  Launched by torchrun (RANK / WORLD_SIZE in
          the environment) or, when WORLD_SIZE is unset, by this script itself: the parent starts N rank processes BEFORE
          anything touches the GPU and never touches it itself.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import glob
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

if "--cpu-child" in sys.argv:   # a helper process of the all-cores CPU baseline (cpu_child_main): pin to its slice of the cores and size the BLAS /
    _k, _P = (int(v) for v in sys.argv[sys.argv.index("--cpu-child") + 1].split("/")[:2])   # OpenMP pools BEFORE numpy loads its OpenBLAS
    _cpus = sorted(os.sched_getaffinity(0))
    _per = max(len(_cpus) // _P, 1)
    _mine = _cpus[_k * _per:(_k + 1) * _per] or _cpus
    os.sched_setaffinity(0, _mine)
    os.environ["OMP_NUM_THREADS"] = str(len(_mine))
    os.environ["OPENBLAS_NUM_THREADS"] = str(min(len(_mine), 64))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_TF = {"f32": 157.3, "bf16": 2500.0, "fp8": 5000.0}   # dense MFMA peaks (same guide; no 2:1 sparsity)
SEED_TABLES, SEED_IDX, SEED_WEIGHTS = 0xF1EE7, 1234, 99
N_IDX_BUFFERS = 1024         # distinct index buffers rotated through (>= 32), so caches are not re-hit artificially.  Round 6: 64 of them
                             # (770 k distinct Model-A rows = 98 MB of lines) sit INSIDE the 256 MB Infinity Cache and read 0.9 % high (71.05 vs 70.4 M
                             # inf/s; the kernel's own time does not move: profiles/r06_headline_index_buffers.txt); 1024 = 1.6 GB of lines do not
STEADY_S = 2.2               # minimum wall clock of the timed region behind `value`
PROFILE_ROUND = "r06"       # prefix of the committed rocprofv3 summaries the roofline objects quote (profiles/<round>_*_kernel_stats.csv)
PMC_FILES = [os.path.join(ROOT, "profiles", n) for n in ("r06_pmc.json", "r05_pmc.json", "archive/r04_pmc.json", "archive/r03_pmc.json", "archive/r02_pmc.json")]   # newest first, entry by entry (see pmc())


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU self-launch (no GPU call may precede this)
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` with no WORLD_SIZE: start N fresh rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set) and wait for them.  The parent has not imported the library or torch.cuda at this point and never does.
    FAIL FAST: the children are polled; the first one that exits non-zero (or dies on a signal) takes the others down within a
    second -- a rank that is gone can never reach the next barrier, and its peers would otherwise sit in the collective until the
    backend's timeout (minutes).  Only processes this function started are ever signalled (their own Popen handles)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "FR_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, set(range(n))
    while live and rc == 0:
        time.sleep(0.05)
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code if code > 0 else 128 - code
                sys.stderr.write("bench.py: rank %d exited with status %d: stopping the other %d rank(s)\n" % (r, code, len(live)))
                break
    if rc != 0:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + 5.0
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc



# ------------------------------------------------------------------------------------------------------------------
# output: ONE compact JSON line on stdout (the driver parses it; VERDICT r03: a 21.7 KB line did not parse), everything else in a file
# ------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
DETAIL_FILE = os.environ.get("FR_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(x, sig=6):
    """Round floats to `sig` significant digits (the line is a summary; the detail file keeps every digit)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_line(result):
    """The line the driver reads: the contract keys + `roofline` + `cpu_baseline` + the few figures the reference's own one-screen
    report would carry (cuda_server.c:565-591).  Everything else of `result` lives in the detail file."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: result.get(k) for k in keys}
    cfg = dict(result.get("config") or {})
    line["config"] = cfg
    for k in ("timed_batches", "timed_s", "value_pcie_inclusive", "value_hbm_resident", "sharded_error", "leg_seconds_total"):
        if result.get(k) is not None:
            line[k] = result[k]
    rf = result.get("roofline")
    if rf:
        line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_name", "avg_launch_ms", "batches_per_launch", "achieved_is",
                                      "concurrent_launches", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "profiled_avg_launch_us", "profile", "pmc_mfma_busy_fraction"))
        line["roofline"].setdefault("traffic", None)
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "gather_only", "fc_only", "end_to_end", "gather_threads", "fc_threads", "host_cpus_usable",
                                          "blas", "sample", "error"))
        if isinstance(cb.get("all_cores"), dict):   # the same sample as P pinned processes on every usable core (detail file: per process)
            line["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("processes", "cores", "end_to_end", "fc_only", "error"))
        for k, n in (("blas", 56), ("sample", 230)):   # the long forms are in the detail file
            if isinstance(line["cpu_baseline"].get(k), str) and len(line["cpu_baseline"][k]) > n:
                line["cpu_baseline"][k] = line["cpu_baseline"][k][:n - 3] + "..."
    for gk in ("gather_per_bank", "gather"):
        g = result.get(gk)
        if g and "achieved" in g:
            line[gk] = _pick(g, ("achieved", "frac", "traffic", "unit", "kernel_name", "avg_launch_ms"))
    rows = {}
    for c in result.get("configs") or []:
        if "tag" in c and c.get("value") is not None:
            rows[c["tag"]] = _r({"inf_per_s": c["value"], "kernel_frac": (c.get("roofline") or {}).get("frac")}, 4)
    if rows:
        line["other_configs"] = rows
    for k in ("gather_per_bank_all_ranks",):
        if result.get(k):
            line[k] = _pick(result[k], ("achieved", "frac", "unit", "ranks_measured"))
    if result.get("configs_all_ranks"):
        line["configs_all_ranks"] = {c.get("tag", str(i)): _r({"inf_per_s": c["value"], "ranks": c.get("ranks_measured")}, 4) for i, c in enumerate(result["configs_all_ranks"])}
    for k in ("sharded", "sharded_inflated_fp8"):
        sh = result.get(k)
        if sh:
            c_ = sh.get("config") or {}
            line[k] = _r({"value": sh.get("value"), "ms_per_step": sh.get("ms_per_step"), "dtype": sh.get("dtype"), "n_gpus": sh.get("n_gpus"), "scaling": sh.get("scaling"),
                          "exchange": c_.get("exchange"),
                          # the leg's own verification: pipelined == stepwise where both forms exist (GPU); on the CPU rehearsal, which has no
                          # pipelined form, the sharded scores against an unsharded context's, bit for bit
                          "ok": bool(c_.get("pipelined_equals_stepwise")) if c_.get("pipelined_equals_stepwise") is not None
                                else (bool(c_["sharded_vs_unsharded_context"].get("bit_identical")) if c_.get("sharded_vs_unsharded_context") else None)}, 5)   # None: nothing was compared
    if result.get("sharded_cabi_rccl"):   # the C-ABI's own RCCL step with G = N ranks (child processes): ok / rate / ranks, or the error
        line["sharded_cabi_rccl"] = _r(_pick(result["sharded_cabi_rccl"], ("ok", "ranks_ok", "world", "inferences_per_s", "ms_per_step", "fp32_max_rel_err_vs_unsharded_first_512", "error")), 5)
        if isinstance(line["sharded_cabi_rccl"].get("error"), str):
            line["sharded_cabi_rccl"]["error"] = line["sharded_cabi_rccl"]["error"][:160]
    line["detail"] = os.path.relpath(DETAIL_FILE, ROOT)
    line = _r(line)
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:   # never let a growing leg list break the parser again: drop the optional summaries, largest first
        for k in ("other_configs", "configs_all_ranks", "gather", "sharded_inflated_fp8", "sharded", "gather_per_bank_all_ranks"):
            line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) < LINE_LIMIT:
                break
    if len(s) >= LINE_LIMIT:   # the mandatory keys alone are too long: shorten the free-text fields, the workload string last
        for holder, key, keep in ((line.get("cpu_baseline") or {}, "sample", 60), (line.get("cpu_baseline") or {}, "blas", 24), (line.get("roofline") or {}, "kernel_name", 48),
                                  (line["config"], "workload", 200), (line["config"], "workload", 60)):
            if isinstance(holder.get(key), str) and len(holder[key]) > keep:
                holder[key] = holder[key][:keep - 3] + "..."
            s = json.dumps(line, separators=(",", ":"))
            if len(s) < LINE_LIMIT:
                break
    assert len(s) < LINE_LIMIT, "bench.py: the stdout line is %d bytes (limit %d)" % (len(s), LINE_LIMIT)
    return s


def emit(result, final=True):
    """Write the full result to the detail file, print the compact line (rank 0 only calls this)."""
    try:
        os.makedirs(os.path.dirname(DETAIL_FILE), exist_ok=True)
        with open(DETAIL_FILE + ".tmp", "w") as f:
            json.dump(result, f, indent=1)
        os.replace(DETAIL_FILE + ".tmp", DETAIL_FILE)
    except OSError as ex:
        sys.stderr.write("bench.py: detail file not written: %r\n" % ex)
    if final:
        sys.stdout.write(compact_line(result) + "\n")
        sys.stdout.flush()


class LineGuard:
    """Legs with data-path collectives inside their step loop: a rank that fails there leaves its peers in a collective.  The headline
    must not be lost to that, and the exit status must say what happened (the reference exits with
    EXIT_FAILURE on socket errors, cuda_server.c:366-398).  While armed, whatever happens -- the leg raises, it exceeds its limit, or
    the launcher terminates this rank because a peer died (SIGTERM; seen through signal.set_wakeup_fd by a helper thread, since the main
    thread may sit inside a collective's C code where Python handlers do not run) -- rank 0 prints the line as it stands with
    `sharded_error`, and the process leaves through os._exit(4).  Never a re-exec: this process has touched the GPU."""
    STATUS = 4

    def __init__(self, rank, result):
        import signal
        import threading
        self.rank, self.result = rank, result
        self.lock = threading.Lock()
        self.done = None
        self.leg = None
        self.rfd, self.wfd = os.pipe()
        os.set_blocking(self.wfd, False)
        self.old_handler = signal.signal(signal.SIGTERM, lambda *_: None)   # a Python-level handler must exist for the wake-up fd to be written
        self.old_fd = signal.set_wakeup_fd(self.wfd, warn_on_full_buffer=False)
        threading.Thread(target=self._watch_signal, daemon=True).start()

    def _watch_signal(self):
        try:
            if os.read(self.rfd, 1):
                self.bail("%s: this rank was terminated by the launcher (a peer rank failed)" % (self.leg or "collective leg"))
        except OSError:
            pass

    def bail(self, why):
        with self.lock:   # first caller wins, and never returns
            if self.rank == 0 and self.result is not None:
                self.result["sharded_error"] = why
                emit(self.result)
            sys.stderr.write("rank %d: %s\n" % (self.rank, why))
            sys.stderr.flush()
            os._exit(self.STATUS)

    def run(self, name, limit, fn):
        import threading
        self.leg = name
        done = threading.Event()
        threading.Thread(target=lambda: None if done.wait(limit) else self.bail("%s did not finish within %.0f s" % (name, limit)), daemon=True).start()
        try:
            return fn()
        except BaseException as ex:   # noqa: BLE001 -- whatever it was, the peers are waiting in a collective: report and leave
            done.set()
            self.bail("%s failed on rank %d: %r" % (name, self.rank, ex))
        finally:
            done.set()

    def close(self):
        import signal
        signal.set_wakeup_fd(self.old_fd)
        signal.signal(signal.SIGTERM, self.old_handler)
        os.close(self.wfd)   # the watcher's read returns b"" and it ends


# ------------------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------------------
def gather_bytes_per_inference(model, fr):
    """row reads + index reads + dense read + record write (index reads follow the index mode)."""
    rows = sum(s.len * 4 for s in model.segments() if s.kind != fr.SEG_DENSE)
    return rows + 4 * model.idx_cols + 4 * model.dense_len + 4 * model.record_len


def fc_flops_per_inference(fc):
    return 2 * sum(fc[i] * fc[i + 1] for i in range(4))


def pmc(key, field=None):
    """Committed PMC summary (tools/pmc_passes.sh -> profiles/archive/r02_pmc.json): separate rocprofv3 --pmc passes of this script's legs,
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md.  -> the entry, one field of it, or None."""
    for path in PMC_FILES:
        try:
            e = json.load(open(path)).get(key)
        except Exception:
            e = None
        if e is not None:
            e = dict(e, pmc_file=os.path.relpath(path, ROOT))
            return e if field is None else e.get(field)
    return None


def find_profile(suffix):
    """Newest committed rocprofv3 summary profiles/rNN_<suffix> (this round's, else an earlier round's for legs whose kernel did not change)."""
    for rnd in ("r06", "r05", "archive/r04", "archive/r03", "archive/r02"):   # (rounds 1-4 live under profiles/archive/)
        if os.path.exists(os.path.join(ROOT, "profiles", "%s_%s" % (rnd, suffix))):
            return "%s_%s" % (rnd, suffix)
    return "%s_%s" % (PROFILE_ROUND, suffix)


def profiled_avg_us(csv_name, kernel):
    """Average duration (us) rocprofv3 --kernel-trace --stats recorded for `kernel` in profiles/<csv_name> (the committed summary of the
    same command): the figure every roofline object quotes beside its live HIP-event time.  None when the file or the kernel is absent."""
    import csv
    path = os.path.join(ROOT, "profiles", csv_name)
    if not kernel or not os.path.exists(path):
        return None
    key = kernel.split("(")[0].strip()
    try:
        for row in csv.DictReader(open(path)):
            name = row.get("Name", "")
            if name.startswith(key) or ("void " + key) in name or key in name:
                return float(row["AverageNs"]) / 1e3
    except Exception:
        return None
    return None


def uniform_idx(rng, ranges, B):
    return (rng.random((B, len(ranges))) * ranges[None, :]).astype(np.int32)


def zipf_idx(rng, rows, B, alpha=1.05):
    u = rng.random((B, len(rows)))
    x = ((rows[None, :].astype(np.float64) ** (1.0 - alpha) - 1.0) * u + 1.0) ** (1.0 / (1.0 - alpha))
    return np.minimum(np.floor(x) - 1, rows[None, :] - 1).astype(np.int32)


def steady_run(run_fn, min_s, n_first=4096, env=None, quantum=256):
    """Time run_fn(n) (-> elapsed seconds of n back-to-back batches) for at least min_s: a calibration run sizes n."""
    el = run_fn(n_first)
    n = max(n_first, int(n_first / max(el, 1e-6) * min_s))
    n = (n + quantum - 1) // quantum * quantum
    if env is not None:
        n = int(env.max_over_ranks(n))
    return n


def bf16_launch_group(B):
    """Batches per launch of the bf16 rows (fr_ctx_set_stream_group; the context's default is 64): 131072 items, at most 256 batches -- Model-B
    batch 1024: 128 (8 tiles per persistent workgroup: 340 M inf/s against 330 M at 64), Model-A batch 256: 256 (489 against 435 M);
    profiles/archive/r03_fused_hs_items_ab.txt, profiles/archive/r03_group_above_64_ab.txt.  Every row says which group it ran at."""
    return max(64, min(256, 131072 // B))


def time_launches(wk, push, per_launch, launches, warm_launches=4, push_launch=None):
    """HIP events on the worker's own stream around `launches` launches (push(i) enqueues one batch; per_launch of them make
    one launch).  -> average launch duration in ms.  With push_launch (launch_pusher below) a launch's batches are pushed by ONE
    native call: a Python push costs ~0.7 us, which is a batch's whole share of a launch for the small-batch low-precision rows
    (Model-A fp8: 64 pushes per 35 us kernel) -- the stream, not the interpreter, must set the pace of a roofline figure."""
    # the launches are timed in blocks and the MEDIAN block counts: one disturbed stretch (another process's burst on the host, a clock
    # step) must not set a roofline figure (seen once: 440 us for a 375 us kernel whose throughput leg in the same run was normal)
    blocks = 5 if launches >= 50 else 1
    per_block = launches // blocks
    if push_launch is not None:
        for l in range(warm_launches):
            push_launch(l)
        wk.sync()
        ms = []
        for b_ in range(blocks):
            wk.timer_start()
            for l in range(per_block):
                push_launch(b_ * per_block + l)
            ms.append(wk.timer_stop_ms() / per_block)
        wk.sync()
        return float(np.median(ms))
    for i in range(warm_launches * per_launch):
        push(i)
    wk.sync()
    ms = []
    for b_ in range(blocks):
        wk.timer_start()
        for i in range(per_block * per_launch):
            push(b_ * per_block * per_launch + i)
        ms.append(wk.timer_stop_ms() / per_block)
    wk.sync()
    return float(np.median(ms))


def launch_pusher(wk, B, d_idx, d_dense, ring, per_launch, n_lists=8):
    """push_launch(l) for time_launches: launch l's per_launch batches through fr_worker_push_device_list, the same buffers in the same
    order as `push(l * per_launch + k)` would take (index buffers rotate through d_idx, score buffers through the two halves of `ring`)."""
    lists = []
    for l in range(n_lists):
        ii = [l * per_launch + k for k in range(per_launch)]
        lists.append(wk.make_push_list([B] * per_launch, [d_idx[i % len(d_idx)] for i in ii],
                                       [d_dense[i % len(d_dense)] for i in ii] if d_dense else None, [ring[i % len(ring)] for i in ii]))
    return lambda l: wk.push_device_list(lists[l % n_lists])


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: memory-resident OpenMP gather + OpenBLAS sgemm chain (checker side; rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------------------------------
def load_blas():
    """BASELINE.md section 2 probe order: a system libopenblas (cblas_sgemm, LP64) -> numpy's bundled libscipy_openblas64_
    (scipy_cblas_sgemm64_, ILP64) -> scipy's bundled libscipy_openblas (scipy_cblas_sgemm, LP64) -> None (torch.mm on CPU)."""
    import ctypes.util
    cands = []
    p = ctypes.util.find_library("openblas")
    if p:
        cands.append((p, "cblas_sgemm", ctypes.c_int, "openblas_get_config", "openblas_get_num_threads"))
    for pat in ("/usr/lib/x86_64-linux-gnu/libopenblas.so*", "/usr/lib64/libopenblas.so*", "/usr/lib/libopenblas.so*", "/opt/*/lib/libopenblas.so*"):
        for f in sorted(glob.glob(pat)):
            cands.append((f, "cblas_sgemm", ctypes.c_int, "openblas_get_config", "openblas_get_num_threads"))
    npd = os.path.join(os.path.dirname(os.path.dirname(np.__file__)), "numpy.libs")
    for f in sorted(glob.glob(os.path.join(npd, "libscipy_openblas64_*.so"))):
        cands.append((f, "scipy_cblas_sgemm64_", ctypes.c_int64, "scipy_openblas_get_config64_", "scipy_openblas_get_num_threads64_"))
    try:
        import scipy
        spd = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        for f in sorted(glob.glob(os.path.join(spd, "libscipy_openblas-*.so"))):
            cands.append((f, "scipy_cblas_sgemm", ctypes.c_int, "scipy_openblas_get_config", "scipy_openblas_get_num_threads"))
    except Exception:
        pass
    for path, sym, itype, cfg, nthr in cands:
        try:
            L = ctypes.CDLL(path)
            fn = getattr(L, sym)
        except Exception:
            continue
        fp = ctypes.POINTER(ctypes.c_float)
        fn.restype = None
        fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, itype, itype, itype, ctypes.c_float, fp, itype, fp, itype, ctypes.c_float, fp, itype]
        ident, threads = os.path.basename(path), None
        try:
            g = getattr(L, cfg)
            g.restype = ctypes.c_char_p
            ident += ": " + g().decode()
            t = getattr(L, nthr)
            t.restype = ctypes.c_int
            threads = int(t())
        except Exception:
            pass
        return {"sgemm": fn, "name": ident, "symbol": sym, "threads": threads}
    return None


def cpu_fc_chain(blas, X, ws, fc, bufs):
    """cuda_server.c:468-491 on the host: R1 = W1*X, R2 = W2*R1, R3 = W3*R2, out = Wout*R3, column-major, alpha = 1, beta = 0.
    X: float32 [B][K] (= column-major K x B); ws[l]: column-major H x K flattened; bufs: pre-allocated R1..R3, out."""
    B = X.shape[0]
    fp = ctypes.POINTER(ctypes.c_float)
    src = X
    for l in range(4):
        H, K = fc[l + 1], fc[l]
        dst = bufs[l]
        if blas is None:
            import torch
            w = torch.from_numpy(ws[l]).view(K, H)                     # element (h, k) at [k][h]
            torch.mm(torch.from_numpy(src).view(B, K), w, out=torch.from_numpy(dst).view(B, H))
        else:   # CblasColMajor = 102, CblasNoTrans = 111: C (H x B, ld H) = A (H x K, ld H) * B (K x B, ld K)
            blas["sgemm"](102, 111, 111, H, B, K, 1.0, ws[l].ctypes.data_as(fp), H, src.ctypes.data_as(fp), K, 0.0, dst.ctypes.data_as(fp), H)
        src = dst
    return bufs[3]


def cpu_child_main(spec):
    """`bench.py --cpu-child K/P/BUDGET`: one of P host processes of the all-cores CPU baseline (leg_cpu_baseline).  Pinned to the K-th slice of
    the usable cores BEFORE any BLAS / OpenMP pool exists, it runs the SAME sample as the parent -- Model-A, 16 384 items per call, bank images in
    host RAM, OpenMP gather + 4 chained sgemm calls -- for BUDGET seconds and prints one JSON line.  Never touches a GPU: the weights come from
    the library's CPU back-end (same content function as the device's)."""
    k, P, budget = spec.split("/")
    k, P, budget = int(k), int(P), float(budget)
    mine = sorted(os.sched_getaffinity(0))   # (pinned at the top of this file, before numpy's OpenBLAS sized its pool)
    import __graft_entry__ as graft
    fr = graft.load_package()
    O = graft.load_oracle()
    O.lib().oracle_set_num_threads(len(mine))
    model = fr.Model.builtin(fr.MODEL_A)
    cctx = fr.Context(model.clone(max_rows=64), device=fr.DEVICE_CPU)       # weights only (the tables used below are the oracle's bank images)
    cctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    ws = [cctx.get_weights(l) for l in range(4)]
    cctx.close()
    om = O.OracleModel("A")
    h = om.halves[0]
    imgs = h.bank_images_native(O.FILL_HASH, SEED_TABLES)
    fc = model.fc
    n = 64 * 256
    rng_c = np.random.default_rng(SEED_IDX + 1000 * (k + 1))
    groups = []
    for _ in range(4):
        src = uniform_idx(rng_c, model.rows(), n)
        hi_ = np.empty_like(src)
        hi_[:, h.wire_to_round] = src
        groups.append(hi_)
    rec = np.empty((n, h.record_len), dtype=np.uint32)
    bufs = [np.empty((n, fc[l + 1]), dtype=np.float32) for l in range(4)]
    blas = load_blas()
    X = rec.view(np.float32)

    def timed(fn):
        fn(0)
        reps, t_begin = 0, time.perf_counter()
        while True:
            fn(reps)
            reps += 1
            el = time.perf_counter() - t_begin
            if el >= budget and reps >= 2:
                return n * reps / el, reps
    h.gather_direct(groups[0], True, imgs, out=rec)
    f_rate, f_reps = timed(lambda i: cpu_fc_chain(blas, X, ws, fc, bufs))
    e_rate, e_reps = timed(lambda i: (h.gather_direct(groups[i % 4], True, imgs, out=rec), cpu_fc_chain(blas, X, ws, fc, bufs)))
    print(json.dumps({"cpu_child": k, "cpus": len(mine), "fc_only": f_rate, "end_to_end": e_rate, "calls": [f_reps, e_reps],
                      "blas_threads": blas["threads"] if blas else None}), flush=True)


def cpu_all_cores(usable, per_proc, budget_s):
    """The same sample on EVERY usable core: P = usable / per_proc processes side by side, each pinned to its own slice (one BLAS build caps at 64
    threads).  -> summed rates, or an {"error": ...} that leaves the single-process figures standing."""
    P = max(usable // max(per_proc, 1), 1)
    if P < 2:
        return None
    try:
        env = dict(os.environ)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-child", "%d/%d/%g" % (k, P, budget_s)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
                 for k in range(P)]
        outs = []
        for pr in procs:
            o, e = pr.communicate(timeout=120)
            if pr.returncode != 0:
                return {"error": "child exited %d: %s" % (pr.returncode, e.decode(errors="replace")[-300:])}
            outs.append(json.loads([ln for ln in o.decode().splitlines() if ln.startswith("{")][-1]))
        return {"processes": P, "threads_per_process": per_proc, "cores": sum(o_["cpus"] for o_ in outs), "fc_only": sum(o_["fc_only"] for o_ in outs),
                "end_to_end": sum(o_["end_to_end"] for o_ in outs), "per_process_end_to_end": [o_["end_to_end"] for o_ in outs],
                "what": "P processes side by side, each pinned to its own slice of the usable cores, each running the same sample (own index rows, same tables / "
                        "weights law) for the same wall time: the rates are summed"}
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def leg_cpu_baseline(graft, ctx, model, idx_host, B, gpu_scores_first, budget_s=2.5):
    """Model-A on the node's host cores: tables materialised in host RAM exactly as host.cpp lays out the card's banks (1.4 GB),
    OpenMP gather that READS them, sgemm chain through OpenBLAS over the same 64 x 256 = 16384-item grouping one GPU launch
    gets.  Three figures: gather-only, FC-only, end-to-end (each >= budget_s of wall clock)."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # idle OpenMP threads must not spin beside OpenBLAS's pool (and vice versa)
    O = graft.load_oracle()
    om = O.OracleModel("A")
    h = om.halves[0]
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    O.lib().oracle_set_num_threads(usable)
    t0 = time.perf_counter()
    imgs = h.bank_images_native(O.FILL_HASH, SEED_TABLES)
    t_fill = time.perf_counter() - t0
    group = 64
    idx = np.concatenate(idx_host[:group], axis=0)                      # [16384][47] in wire order: the GPU run's own 64 index buffers
    n = idx.shape[0]
    # ... plus 7 more groups of the same index law, rotated through: 8 x 770 k distinct rows = ~400 MB of cache lines, so that the
    # gather reads MEMORY instead of re-hitting one group's rows in the host's last-level cache
    rng_c = np.random.default_rng(SEED_IDX + 77)
    n_groups = 8
    hidx_groups = []
    for gi in range(n_groups):
        src = idx if gi == 0 else uniform_idx(rng_c, model.rows(), n)
        hi_ = np.empty_like(src)
        hi_[:, h.wire_to_round] = src                                    # wire order -> (bank, round) order
        hidx_groups.append(hi_)
    hidx = hidx_groups[0]
    ws = [ctx.get_weights(l) for l in range(4)]
    fc = model.fc
    rec = np.empty((n, h.record_len), dtype=np.uint32)
    bufs = [np.empty((n, fc[l + 1]), dtype=np.float32) for l in range(4)]
    blas = load_blas()
    # gather threads: the fastest of {all usable cores, 1/2, 1/4, 64, 32} on a short probe (hyper-threads and cgroup CPU quotas make
    # "all" a poor choice on some hosts); `cores` reports what the timed runs actually used
    probe = {}
    for nt in sorted({usable, max(usable // 2, 1), max(usable // 4, 1), min(64, usable), min(32, usable)}, reverse=True):
        O.lib().oracle_set_num_threads(nt)
        h.gather_direct(hidx, True, imgs, out=rec)
        t1 = time.perf_counter()
        h.gather_direct(hidx, True, imgs, out=rec)
        probe[nt] = time.perf_counter() - t1
    threads = min(probe, key=probe.get)
    O.lib().oracle_set_num_threads(threads)

    def timed(fn):
        fn(0)
        reps, t_begin = 0, time.perf_counter()
        while True:
            fn(reps)
            reps += 1
            el = time.perf_counter() - t_begin
            if el >= budget_s and reps >= 2:
                return n * reps / el, reps

    g_rate, g_reps = timed(lambda i: h.gather_direct(hidx_groups[i % n_groups], True, imgs, out=rec))
    X = rec.view(np.float32)
    f_rate, f_reps = timed(lambda i: cpu_fc_chain(blas, X, ws, fc, bufs))
    # second FC engine: torch.mm on the host (its own BLAS and thread pool -- numpy's bundled OpenBLAS is built for <= 64 threads); the
    # faster of the two runs the end-to-end figure, both are reported
    alt = None
    if blas is not None:
        try:
            import torch
            budget_main, budget_s = budget_s, min(budget_s, 1.0)
            # torch.mm at its own default thread count AND at every usable core (the baseline must not idle three quarters
            # of the host because one BLAS build caps at 64 threads); the faster setting stays and is what `cores` reports
            t_default = torch.get_num_threads()
            tries = {}
            for nt in sorted({t_default, usable}):
                torch.set_num_threads(nt)
                tries[nt] = timed(lambda i: cpu_fc_chain(None, X, ws, fc, bufs))
            t_threads = max(tries, key=lambda k: tries[k][0])
            torch.set_num_threads(t_threads)
            t_rate, t_reps = tries[t_threads]
            budget_s = budget_main
            alt = {"engine": "torch.mm", "fc_only": t_rate, "fc_GFLOPs": t_rate * fc_flops_per_inference(fc) / 1e9, "threads": t_threads,
                   "fc_only_by_threads": {str(k): v[0] for k, v in tries.items()},
                   "build": ", ".join(tok.strip() for ln in torch.__config__.show().split("\n") for tok in ln.split(",") if "BLAS_INFO" in tok or "USE_MKL=" in tok),
                   "calls": t_reps}
            if t_rate > f_rate:
                alt["openblas_fc_only"] = f_rate
                f_rate, f_reps, blas_used = t_rate, t_reps, None
            else:
                blas_used = blas
        except Exception as ex:
            alt, blas_used = {"error": repr(ex)[:200]}, blas
    else:
        blas_used = blas
    e_rate, e_reps = timed(lambda i: (h.gather_direct(hidx_groups[i % n_groups], True, imgs, out=rec), cpu_fc_chain(blas_used, X, ws, fc, bufs)))
    h.gather_direct(hidx_groups[0], True, imgs, out=rec)               # group 0 = the GPU's index buffers: cross-check the first batch
    cpu_fc_chain(blas_used, X, ws, fc, bufs)
    cpu_scores = bufs[3].ravel()[:B].copy()
    err = float(np.abs(cpu_scores - gpu_scores_first).max() / max(np.abs(cpu_scores).max(), 1e-30)) if gpu_scores_first is not None else None
    gbytes = 1408 + 188 + 1408
    fc_threads = (alt or {}).get("threads") if blas_used is None else blas["threads"]
    # ... and on EVERY usable core: the single-process figure above leaves three quarters of a 256-thread host idle in its FC phase (one OpenBLAS
    # build = 64 threads); P pinned processes side by side do not.  `value` = the better of the two, `cores` = what that one used.
    # (Measured on the GPU boxes of this pool: 4 x 64 pinned threads run the sample at a THIRD of the one 64-thread process's rate -- the 256 "usable" CPUs
    # are a scheduler view, the box's CPU share is far smaller -- so the single process stays the baseline there; the line carries both.)
    allc = cpu_all_cores(usable, max(fc_threads or 64, 1), min(budget_s, 1.0)) if os.environ.get("FR_BENCH_CPU_ALL_CORES", "1") != "0" else None
    # `cores` = the most host threads any phase of the end-to-end figure ran on: the gather's OpenMP team (`gather_threads`, the fastest of the
    # probe above -- all cores is not the fastest: hyper-threads / cgroup quotas) and the FC engine's pool (`fc_threads`: OpenBLAS builds cap
    # at 64, torch.mm takes its own default) run one after the other, never at the same time
    best_all = bool(allc and allc.get("end_to_end", 0) > e_rate)
    return {"value": allc["end_to_end"] if best_all else e_rate, "unit": "inferences/s", "cores": allc["cores"] if best_all else max(threads, fc_threads or 0),
            "gather_threads": threads, "fc_threads": fc_threads, "all_cores": allc, "single_process_end_to_end": e_rate,
            "kind": "port",
            "gather_only": g_rate, "fc_only": f_rate, "end_to_end": e_rate,
            "gather_GBps_algorithmic": g_rate * gbytes / 1e9, "fc_GFLOPs": f_rate * fc_flops_per_inference(fc) / 1e9,
            "blas": blas["name"] if blas else "torch.mm (" + __import__("torch").__config__.parallel_info().split("\n")[0] + ")",
            "blas_symbol": blas["symbol"] if blas else "torch.mm", "blas_threads": blas["threads"] if blas else None,
            "fc_engine_of_end_to_end": "torch.mm" if blas_used is None else blas_used["symbol"], "fc_second_engine": alt,
            "gather_thread_probe_s": {str(k): v for k, v in probe.items()}, "host_cpus_usable": usable,
            "gpu_vs_cpu_max_rel_err_first_batch": err, "host_table_bytes": int(sum(im.nbytes for im in imgs)), "host_table_fill_s": t_fill,
            "sample": "Model-A, %d items per call (64 batches of 256 = one fused GPU launch), gather-only %d calls, FC-only %d, end-to-end %d (>= %.1f s each); "
                      "same seeded tables / weights / index law, 8 index groups rotated (the first = the GPU run's buffers); gather = OpenMP over items reading "
                      "bank images in host RAM (oracle_gather_banks_direct), FC = 4 chained column-major sgemm calls; `all_cores` = the same sample as P processes "
                      "side by side on every usable core, `value` = the better of the two" % (n, g_reps, f_reps, e_reps, budget_s)}


def leg_cpu_backend(fr, idx_host, B, gpu_scores_first):
    """BASELINE configs[0] -- "Model-A (smallest user_krnl config), batch=1 ... on host CPU (plumbing, no accelerator)" -- through the PRODUCT:
    the library's own CPU back-end (fr_ctx_create(model, device = -1): csrc/fr_cpu.cpp, the same C-ABI, no checker code involved).
    Full-size Model-A in host memory; one request of one item through fr_worker_submit + fr_worker_sync (median of 300), and the 16384-item
    sample of the cpu_baseline leg through the same entry points on every usable core."""
    m = fr.Model.builtin(fr.MODEL_A)
    t0 = time.perf_counter()
    threads = fr.cpu_set_threads(0)
    ctx = fr.Context(m, device=fr.DEVICE_CPU)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    t_setup = time.perf_counter() - t0
    w1 = fr.Worker(ctx, 1)
    one = idx_host[0][:1]
    for _ in range(30):
        w1.infer(one)
    ts = []
    for _ in range(300):
        t1 = time.perf_counter()
        s1 = w1.infer(one)
        ts.append(time.perf_counter() - t1)
    ts.sort()
    w1.close()
    big = np.concatenate(idx_host[:64], axis=0)
    wk = fr.Worker(ctx, big.shape[0])
    sc = wk.infer(big)
    reps, t_begin = 0, time.perf_counter()
    while reps < 2 or time.perf_counter() - t_begin < 1.5:
        wk.infer(big)
        reps += 1
    rate = big.shape[0] * reps / (time.perf_counter() - t_begin)
    wk.close()
    ctx.close()
    err = float(np.abs(sc[:B] - gpu_scores_first).max() / max(np.abs(gpu_scores_first).max(), 1e-30)) if gpu_scores_first is not None else None
    return {"batch_1": {"us_p50": 1e6 * ts[len(ts) // 2], "us_p90": 1e6 * ts[9 * len(ts) // 10], "inferences_per_s": 1.0 / ts[len(ts) // 2],
                        "score_equals_batch_of_16384": bool(s1[0] == sc[0]),
                        "what": "BASELINE configs[0]: Model-A batch 1 on the host CPU through the library's CPU back-end (fr_ctx_create device = -1; fr_worker_submit + "
                                "fr_worker_sync, the ctypes call included): a request of one item runs on one core"},
            "end_to_end": rate, "unit": "inferences/s", "threads": threads, "sample_items": int(big.shape[0]), "calls": reps, "setup_s": t_setup,
            "cpu_vs_gpu_max_rel_err_first_batch": err,
            "what": "the cpu_baseline leg's 16384-item sample (64 batches of 256) through the same back-end: gather over the record-word descriptors + "
                    "k-ordered fp32 multiply-add chain (own code, no BLAS), %d threads" % threads}


# ------------------------------------------------------------------------------------------------------------------
# the request path over TCP: fleetrec_sender -> fleetrec_server --stream on loopback, own processes
# ------------------------------------------------------------------------------------------------------------------
def free_port_block(n):
    import random
    for _ in range(200):
        base = random.randint(20000, 60000 - n)
        socks = []
        try:
            for i in range(n):
                sk = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                sk.bind(("0.0.0.0", base + i))
                socks.append(sk)
            return base
        except OSError:
            continue
        finally:
            for sk in socks:
                sk.close()
    raise RuntimeError("no free block of %d ports" % n)


def leg_tcp_reply(B, device, threads=4):
    """The same path as a SERVICE: fleetrec_server --stream --reply sends the scores back over the socket block by block (adaptive batching,
    small blocks through the stage pipeline), fleetrec_sender --reply --window W times request sent -> scores received.  Three operating points:
    saturated (256 requests in flight per connection), a few in flight (4 per connection) and light (one request per 500 us per connection)."""
    import re
    host = os.path.join(ROOT, "gpu-fpga-recommendation-system_amd", "host")
    srv_bin, snd_bin = os.path.join(host, "fleetrec_server"), os.path.join(host, "fleetrec_sender")
    if not (os.path.exists(srv_bin) and os.path.exists(snd_bin)):
        return {"skipped": "host programs not built (make -C gpu-fpga-recommendation-system_amd/host)"}
    out = {}
    for name, total, window, interval in (("saturated", 400000, 256, 0), ("four_in_flight", 40000, 4, 0), ("light_load", 6000, 256, 500)):
        port = free_port_block(threads)
        common = ["--model", "A", "--batch", str(B), "--threads", str(threads), "--port", str(port)]
        srv = subprocess.Popen([srv_bin] + common + ["--total", str(total), "--device", str(device), "--tables", "hash", "--weights", "uniform", "--stream", "--reply"],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        snd = None
        try:
            time.sleep(0.3)
            snd = subprocess.Popen([snd_bin] + common + ["--indices", "uniform", "--reply", "--window", str(window), "--interval-us", str(interval)],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            so, _ = srv.communicate(timeout=120)
            no, _ = snd.communicate(timeout=30)
        finally:
            for p_ in (srv, snd):
                if p_ is not None and p_.poll() is None:
                    p_.kill()
                    p_.wait()
        so, no = so.decode(errors="replace"), no.decode(errors="replace")
        m1 = re.search(r"first connection -> last scores: ([0-9.]+) s = ([0-9.]+) M inferences/s over TCP", so)
        m2 = re.search(r"latency request sent -> scores received  n=(\d+) avg ([0-9.]+) us  p50 ([0-9.]+)  p90 ([0-9.]+)  p99 ([0-9.]+)", no)
        if srv.returncode != 0 or not (m1 and m2):
            out[name] = {"error": (so + no)[-300:]}
            continue
        out[name] = {"inferences_per_s": float(m1.group(2)) * 1e6, "requests_timed": int(m2.group(1)), "request_to_reply_us_p50": float(m2.group(3)),
                     "request_to_reply_us_p90": float(m2.group(4)), "request_to_reply_us_p99": float(m2.group(5)),
                     "offered": "%d requests in flight per connection" % window if interval == 0 else "one request per %d us per connection" % interval}
    out["what"] = ("fleetrec_sender --reply --window W -> fleetrec_server --stream --reply over loopback TCP, %d connections, batch %d: scores sent back over the "
                   "socket; latency measured at the sender (request sent -> its scores received), first 5 %% of every connection dropped" % (threads, B))
    return out


def leg_tcp(B, device, threads=4, total=1000000):
    """Model-A batch B through the request path the reference has: `threads` TCP connections (the reference's THREAD_NUM = 4 on PORT+i),
    fixed-size blocks of B x 47 int32 indices on the wire, the server's connection threads handing every block to fr_worker_push_host,
    scores delivered to host memory.  The server (its own process and context on the same GPU) reports the rate from its first accept()
    to the join of its threads."""
    import re
    host = os.path.join(ROOT, "gpu-fpga-recommendation-system_amd", "host")
    srv_bin, snd_bin = os.path.join(host, "fleetrec_server"), os.path.join(host, "fleetrec_sender")
    if not (os.path.exists(srv_bin) and os.path.exists(snd_bin)):
        return {"skipped": "host programs not built (make -C gpu-fpga-recommendation-system_amd/host)"}
    port = free_port_block(threads)
    common = ["--model", "A", "--batch", str(B), "--threads", str(threads), "--port", str(port)]
    srv = subprocess.Popen([srv_bin] + common + ["--total", str(total), "--device", str(device), "--tables", "hash", "--weights", "uniform", "--stream"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    snd = None
    try:
        time.sleep(0.3)
        snd = subprocess.Popen([snd_bin] + common + ["--indices", "uniform"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out, _ = srv.communicate(timeout=120)
    finally:
        for p_ in (srv, snd):
            if p_ is not None and p_.poll() is None:
                p_.kill()
                p_.wait()
    out = out.decode(errors="replace")
    m_ = re.search(r"first connection -> last scores: ([0-9.]+) s = ([0-9.]+) M inferences/s over TCP", out)
    if srv.returncode != 0 or not m_:
        return {"error": out[-400:]}
    return {"value": float(m_.group(2)) * 1e6, "unit": "inferences/s", "timed_s": float(m_.group(1)), "timed_batches": total, "connections": threads,
            "what": "fleetrec_sender -> fleetrec_server --stream over loopback TCP (own processes): %d connections, B x 47 int32 per batch on the wire "
                    "(32 pre-drawn uniform blocks per connection in rotation), fr_worker_push_host per block, scores to host memory; the server's clock "
                    "from its first accept() to the join of its connection threads" % threads}


# ------------------------------------------------------------------------------------------------------------------
# GPU legs
# ------------------------------------------------------------------------------------------------------------------
def leg_group_table(fr, ctx, model, B, d_idx, threads, depth):
    """Model-A batch 256 fp32: throughput and latency against the launch group (fr_ctx_set_stream_group): groups below 12 ride the stage
    pipeline (one launch per push, five stages in flight), larger ones the fused item-tile kernels (one launch per group)."""
    out = []
    flops = fc_flops_per_inference(model.fc) * B
    for g in (1, 8, 16, 32, 64):
        ctx.set_stream_group(g)
        dv = fr.Driver(ctx, threads, depth, B)
        dv.run_resident(B, 2048, d_idx)
        n = steady_run(lambda k: dv.run_resident(B, k, d_idx), 0.5, n_first=4096)
        el = dv.run_resident(B, n, d_idx)
        dv.close()
        wk = fr.Worker(ctx, B)
        ring = [fr.DeviceBuffer(ctx, B * 4) for _ in range(2 * g)]
        push = lambda i: wk.push_device(B, d_idx[i % len(d_idx)], None, ring[i % len(ring)])
        ms = time_launches(wk, push, g, 200 if g > 1 else 1000, warm_launches=50, push_launch=launch_pusher(wk, B, d_idx, None, ring, g))
        lat = []
        for _ in range(40):                      # idle worker: first push -> all g batches' scores complete (queueing + launch + sync)
            t0 = time.perf_counter()
            for i in range(g):
                push(i)
            wk.sync()
            lat.append(time.perf_counter() - t0)
        wk.close()
        for b_ in ring:
            b_.free()
        out.append({"group": g, "path": "stage pipeline (one launch per push)" if g < 12 else "fused item-tile kernel (one launch per group)",
                    "inferences_per_s": n * B / el, "launch_ms_one_stream": ms, "tflops_one_stream": flops * g / (ms * 1e-3) / 1e12,
                    "push_to_scores_ms_p50": 1e3 * float(np.median(lat)), "push_to_scores_ms_p90": 1e3 * float(np.percentile(lat, 90))})
    ctx.set_stream_group(64)
    return out


def gemm_workgroups(kname, N, ldm):
    """Workgroups of one launch of a GEMM-layer kernel, from the name the library reports (fr_worker_last_kernel): fc_lp_gemm_kernel<P, MU, GN, ...>
    has GN (n) x 128 MU (m) tiles, fc_gemm_pipe_kernel 128 x 256, fc_pp_gemm_kernel 256 x 256; None for any other kernel (stage bodies: one launch fills the chip)."""
    m_ = re.match(r"fc_lp_gemm_kernel<\d+, (\d+), (\d+)", kname or "")
    if m_:
        mu, gn = int(m_.group(1)), int(m_.group(2))
        return max(1, (N // gn) * (ldm // (128 * mu)))
    if (kname or "").startswith("fc_gemm_pipe_kernel"):
        return max(1, (N // 128) * (ldm // 256))
    if (kname or "").startswith("fc_pp_gemm_kernel"):      # 256 x 256 tiles
        return max(1, (N // 256) * (ldm // 256))
    if (kname or "").startswith("fc_pp_gemm_n128_kernel"):  # 128 x 256 tiles
        return max(1, (N // 128) * (ldm // 256))
    return None


def leg_config(fr, ctx, model, B, precision, d_idx, d_dense, idx_host0, dense_host0, threads, depth, label, min_s=1.0, pmc_key=None, env=None, profile_csv=None, tag=None, roofline=True):
    """One non-headline BASELINE configuration: steady-state throughput (>= 1 s) + an in-run roofline object for its dominant
    kernel from HIP events on one worker's stream."""
    prec_enum = {"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[precision]
    ctx.set_fc_precision(prec_enum)
    if precision == "fp8":
        cal = fr.Worker(ctx, B)
        cal.calibrate_fp8(idx_host0, dense_host0)
        cal.close()
    fc = model.fc
    flops_inf = fc_flops_per_inference(fc)
    # a chain model's GEMM tiles follow the context's chain width (fr_ctx_set_chain_width): decided HERE, on purpose, as a host that drives
    # threads x depth streams would -- not by whichever worker happens to launch first (--roofline-only has no driver)
    ctx.set_chain_width(min(4, threads * depth))
    if min_s > 0:
        dv = fr.Driver(ctx, threads, depth, B)
        dv.run_resident(B, 256, d_idx, d_dense)
        n = steady_run(lambda k: dv.run_resident(B, k, d_idx, d_dense), min_s, n_first=512 if min_s >= 1.0 else 64, quantum=64, env=env)
        world = env.world if env is not None else 1
        if world > 1:   # replicas: every rank times the SAME number of batches between barriers; the slowest rank's time counts
            env.barrier()
            ctx.synchronize()
            t0 = time.perf_counter()
            dv.run_resident(B, n, d_idx, d_dense)
            env.barrier()
            el = env.max_over_ranks(time.perf_counter() - t0)
        else:
            el = dv.run_resident(B, n, d_idx, d_dense)
        dv.close()
        res = {"tag": tag, "workload": label, "dtype": precision, "value": world * n * B / el, "unit": "inferences/s", "timed_batches": n, "timed_s": el,
               "ms_per_step": 1e3 * el / n, "fc_tflops_end_to_end": flops_inf * B * n * world / el / 1e12,
               "frac_of_mfma_peak_end_to_end": flops_inf * B * n / el / 1e12 / MFMA_PEAK_TF[precision]}
    else:   # --roofline-only: no multi-stream loop
        res = {"tag": tag, "workload": label, "dtype": precision, "value": None, "unit": "inferences/s", "timed_batches": 0, "timed_s": 0.0, "ms_per_step": None,
               "fc_tflops_end_to_end": None, "frac_of_mfma_peak_end_to_end": None}
    if not roofline:   # --throughput-only: a profiled run of the multi-stream loop alone (profiles/*_4streams_kernel_stats.csv)
        return res
    wk = fr.Worker(ctx, B)
    group = ctx.stream_group()
    if group > 1:   # fused item-tile kernel: one launch = the whole hot path of min(group, items per launch / B) queued batches
        per_launch = max(1, min(group, (262144 if precision == "bf16" else 16384) // B))   # bf16: the persistent kernel's launches carry as many items as the group allows
        ring = [fr.DeviceBuffer(ctx, B * 4) for _ in range(2 * per_launch)]
        push = lambda i: wk.push_device(B, d_idx[i % len(d_idx)], d_dense[i % len(d_dense)] if d_dense else None, ring[i % len(ring)])
        # (--roofline-only = a PROFILED run: 500 timed launches, so that the 30 cold ones at the head of the process are 6 % of the calls the
        # stats file averages instead of 23 % -- its AverageNs then sits within ~1 % of the steady launch time)
        ms = time_launches(wk, push, per_launch, 500 if min_s == 0 else 100, warm_launches=30, push_launch=launch_pusher(wk, B, d_idx, d_dense, ring, per_launch))
        flops = flops_inf * B * per_launch
        kname = wk.last_kernel()   # the kernel that carried these launches, as the library reports it (fr_worker_last_kernel)
        what = "%s: one launch = gather + 4-GEMM chain of %d queued batches of %d, back-to-back on ONE stream" % (kname, per_launch, B)
        for b_ in ring:
            b_.free()
    else:           # stage pipeline: FC1 runs as its own LDS-tiled GEMM launch (fc_lp_gemm_kernel) -- the dominant kernel
        # measured with the context in the timed region's state -- chain width W = min(threads x depth, 4): the bf16 / fp8 layers take tiles
        # that cover 1 / W of the chip -- and as many workers launching the layer side by side as FIT on the chip together
        # (256 compute units / workgroups per launch): every launch then runs from the moment it is dequeued, so the HIP-event time per
        # launch on its stream is the kernel duration rocprofv3 averages, and the chip's rate is `fit` launches' FLOPs over it.
        d_sc = fr.DeviceBuffer(ctx, B * 4)
        wk.submit_device(B, d_idx[0], d_dense[0] if d_dense else None, d_sc)
        wk.sync()
        side = [wk] + [fr.Worker(ctx, B) for _ in range(threads * depth - 1)]
        for v in side[1:]:   # every worker's activation image is written once (the layers read it)
            v.submit_device(B, d_idx[0], d_dense[0] if d_dense else None, d_sc)
            v.sync()
        layer_ms, layer_kernels, layer_conc, layer_spread = [], [], [], []
        reps = 200   # per stream; the block's fixed cost (first dispatch, stop event: ~0.1 ms under rocprofv3) stays below 1 % of a 47 us launch
        for layer in range(4):
            if layer == 3 and layer_kernels[2].startswith("fc_lp_gemm_out_kernel"):   # the output layer rides in FC3's epilogue: no launch of its own in the chain
                res["output_layer"] = "folded into FC3's epilogue (%s): no launch of its own" % layer_kernels[2]
                break
            wk.fc_layer_only(B, layer)
            layer_kernels.append(wk.last_kernel())
            wk.sync()
            wgs = gemm_workgroups(layer_kernels[-1], fc[layer + 1] if layer < 3 else 1, B)
            act = side[:max(1, min(len(side), 256 // wgs))] if wgs else side[:1]
            for _ in range(10):
                for v in act:
                    v.fc_layer_only(B, layer)
            for v in act:
                v.sync()
            # one host thread per worker (as a server has): start event, `reps` launches from ONE native call (ctypes drops the GIL: the threads
            # really issue side by side, and no Python-level call sits between two launches -- under rocprofv3 such a call costs more than a
            # 57 us kernel takes, and the HIP-event figure of a profiled run read 13-17 % above its own trace), stop event, wait
            stops = [None] * len(act)

            def run_one(i, v):
                v.timer_start()
                v.fc_layer_repeat(B, layer, reps)
                stops[i] = v.timer_stop_ms()
            th = [threading.Thread(target=run_one, args=(i, v)) for i, v in enumerate(act)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            layer_ms.append(float(np.mean(stops)) / reps)        # mean duration of a launch (what the profiler averages)
            layer_conc.append(float(np.sum(stops) / np.max(stops)))  # launches in flight on average: sum of the streams' busy times over the wall time
            layer_spread.append([round(float(x), 3) for x in stops])
        for v in side[1:]:
            v.close()
        ms = layer_ms[0]
        flops = 2 * fc[0] * fc[1] * B
        kname = layer_kernels[0]
        what = ("%s (FC1: %d x %d x %d): the tile chain width %d gives it, as many launches side by side as fit on the chip; avg_launch_ms = mean "
                "launch duration on its stream, concurrent_launches = the streams' busy time over the wall time, achieved = concurrent_launches x "
                "FLOPs per launch / avg_launch_ms = all FLOPs over the wall time" % (kname, fc[0], fc[1], B, ctx.chain_width()))
        res["chain_width"] = ctx.chain_width()
        res["layer_launch_ms"] = layer_ms
        res["layer_kernels"] = layer_kernels
        res["concurrent_launches"] = layer_conc[0]
        res["layer_concurrency"] = layer_conc
        res["layer_stream_busy_ms"] = layer_spread   # per layer: each worker stream's time for its `reps` launches (equal = the streams share the chip fairly)
        d_sc.free()
    wk.close()
    ach = res.get("concurrent_launches", 1) * flops / (ms * 1e-3) / 1e12
    pm = (pmc(pmc_key) or {}) if pmc_key else {}
    res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TF[precision], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TF[precision],
                       "traffic": pm.get("traffic_bytes_per_launch"), "kernel": what, "kernel_name": kname, "avg_launch_ms": ms, "algorithmic_flops_per_launch": flops}
    # what `achieved` is (stage-pipeline rows changed their definition in round 4; the marker makes rows comparable across rounds)
    res["roofline"]["achieved_is"] = "one_launch_alone_on_one_stream"
    if res.get("concurrent_launches"):
        res["roofline"]["concurrent_launches"] = res["concurrent_launches"]   # achieved = concurrent_launches x algorithmic_flops_per_launch / avg_launch_ms
        res["roofline"]["achieved_is"] = "concurrent_launches_side_by_side_x_flops_per_launch_over_avg_launch_ms"
    if profile_csv and os.path.exists(os.path.join(ROOT, "profiles", profile_csv)):   # the committed rocprofv3 --kernel-trace --stats summary of this leg's own command: must agree with avg_launch_ms
        res["roofline"]["profiled_avg_launch_us"] = profiled_avg_us(profile_csv, kname)
        res["roofline"]["profile"] = "profiles/" + profile_csv
        # the same kernel under the row's own MULTI-stream throughput run (as the headline carries it): launches of several streams share the chip,
        # so a launch is longer but more of them are in flight -- which is why a row's end-to-end rate can exceed its one-stream kernel fraction
        try:   # ... and the same run's launch period from its own trace (profiles/rNN_roofline_pairs.json): what the HIP-event figure above measures
            rnd_, leg_ = profile_csv.split("_", 1)[0], profile_csv.split("_", 1)[1].replace("_kernel_stats.csv", "")
            pe = json.load(open(os.path.join(ROOT, "profiles", rnd_ + "_roofline_pairs.json"))).get(leg_) or {}
            if pe.get("rocprofv3_span_per_launch_us"):
                res["roofline"]["profiled_span_per_launch_us"] = pe["rocprofv3_span_per_launch_us"]
                res["roofline"]["profiled_median_us"] = pe.get("rocprofv3_median_us")
        except Exception:
            pass
        ms_csv = profile_csv.replace("_kernel_stats.csv", "_4streams_kernel_stats.csv")
        if os.path.exists(os.path.join(ROOT, "profiles", ms_csv)):
            res["roofline"]["profiled_avg_launch_us_4streams"] = profiled_avg_us(ms_csv, kname)
            res["roofline"]["profile_4streams"] = "profiles/" + ms_csv
    if pm:
        res["roofline"]["pmc_mfma_busy_fraction"] = pm.get("mfma_busy_fraction")
        res["roofline"]["traffic_source"] = "%s[%s]: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x2 gfx950 correction), bytes per launch" % (pm.get("pmc_file"), pmc_key)
    return res


def leg_gather(fr, ctx, model, B, law, reps=200, nbuf=32, seed=SEED_IDX, variants=False):
    """fr_worker_gather_only (record-producing gather, the section-8(d) roofline kernel) at Model-C batch 4096: `nbuf` rotating
    index buffers, `reps` timed launches, HIP events on the worker's stream.  variants=True adds the A/B of the three gather
    kernels on the same index buffers (word-major / item-tile / item-tile + wave-level dedup) with the dedup kernel's own count of
    merged lookups."""
    rng = np.random.default_rng(seed)
    ranges = model.index_ranges()
    mk = (lambda: zipf_idx(rng, ranges, B)) if law == "zipf" else (lambda: uniform_idx(rng, ranges, B))
    host = [mk() for _ in range(nbuf)]
    idxs = [fr.DeviceBuffer.from_numpy(ctx, a) for a in host]
    dns = [fr.DeviceBuffer.from_numpy(ctx, rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32)) for _ in range(nbuf)] if model.dense_len else None
    wk = fr.Worker(ctx, B)
    rec = wk.records_dptr()
    for i in range(20):
        wk.gather_only(B, idxs[i % nbuf], dns[i % nbuf] if dns else None, rec)
    wk.sync()
    blk = []   # five blocks, the median counts (see time_launches)
    for b_ in range(5):
        wk.timer_start()
        for i in range(reps // 5):
            wk.gather_only(B, idxs[(b_ * (reps // 5) + i) % nbuf], dns[(b_ * (reps // 5) + i) % nbuf] if dns else None, rec)
        blk.append(wk.timer_stop_ms() / (reps // 5))
    ms = float(np.median(blk))
    wk.sync()
    kernel_name = wk.last_kernel()   # what fr_worker_gather_only launched (fr_worker_last_kernel)
    ab = None
    if variants:
        ab = {}
        lookups = B * model.n_tables if model.desc.index_mode == fr.INDEX_PER_TABLE else B * model.idx_cols
        for name, var in (("word_major", fr.GATHER_WORD_MAJOR), ("item_tile", fr.GATHER_ITEM_TILE), ("item_tile_dedup", fr.GATHER_ITEM_TILE_DEDUP)):
            ctx.set_gather_variant(var)
            for i in range(20):
                wk.gather_only(B, idxs[i % nbuf], dns[i % nbuf] if dns else None, rec)
            wk.sync()
            wk.timer_start()
            for i in range(reps):
                wk.gather_only(B, idxs[i % nbuf], dns[i % nbuf] if dns else None, rec)
            ab[name] = {"avg_launch_ms": wk.timer_stop_ms() / reps}
            wk.sync()
        ctx.set_gather_variant(fr.GATHER_ITEM_TILE_DEDUP_COUNT)   # un-timed: how many lookups the waves actually merged
        ctx.gather_merged_lookups(reset=True)
        for i in range(nbuf):
            wk.gather_only(B, idxs[i], dns[i] if dns else None, rec)
        wk.sync()
        ab["item_tile_dedup"]["merged_lookup_fraction"] = ctx.gather_merged_lookups() / float(nbuf * lookups)
        ctx.set_gather_variant(fr.GATHER_WORD_MAJOR)
    wk.close()
    for b_ in idxs + (dns or []):
        b_.free()
    gb = gather_bytes_per_inference(model, fr) * B
    out = {"kernel": kernel_name, "kernel_name": kernel_name, "avg_launch_ms": ms, "achieved": gb / (ms * 1e-3) / 1e9, "frac": gb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": gb, "inferences_per_s": B / (ms * 1e-3), "index_buffers": nbuf, "timed_launches": reps}
    if ab is not None:
        out["kernel_ab"] = ab
    if law == "zipf":
        cols = range(0, host[0].shape[1], 7)
        out["duplicate_fraction_within_batch"] = float(np.mean([1.0 - len(np.unique(z[:, t])) / B for z in host[:2] for t in cols]))
    return out


# ------------------------------------------------------------------------------------------------------------------
# sharded mode (BASELINE configs[3] / [4])
# ------------------------------------------------------------------------------------------------------------------
def cabi_child_main(spec):
    """`bench.py --cabi-child RANK/WORLD/DEVICE/IDHEX/OUT`: one rank of the C-ABI's OWN table-sharded step over RCCL (fr_comm_init_rank +
    fr_worker_submit_sharded, csrc/fr_comm.cpp) -- the path no one-GPU box can run with G > 1.  Started by every rank of an N > 1 bench run
    as a child process of its own (leg_cabi_rccl): whatever happens here -- an RCCL set-up that hangs, a failed step -- stays in the child,
    the parent's line is never at stake.  Model-C, batch 4096: fp32 steps checked on rank 0 against an unsharded full-size context
    (first 512 items, 1e-5), then bf16 steps timed.  Writes one JSON object to OUT."""
    r, G, dev, idhex, out_path = spec.split("/", 4)
    r, G, dev = int(r), int(G), int(dev)
    import __graft_entry__ as graft
    fr = graft.load_package()
    res = {"rank": r, "world": G}
    try:
        B = 4096
        m = fr.Model.builtin(fr.MODEL_C)
        ctx = fr.Context(m, device=dev, shard_rank=r, n_shards=G)
        ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
        ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
        wk = fr.Worker(ctx, B)
        comm = fr.Comm.init_rank(ctx, bytes.fromhex(idhex))
        comm.set_wait_ms(20000)
        rng = np.random.default_rng(SEED_IDX)                       # every rank draws the SAME request batches
        reqs = [(uniform_idx(rng, m.rows(), B), rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)) for _ in range(4)]
        first = wk.infer_sharded(comm, *reqs[0])                     # fp32
        res["fp32_scores_sha1"] = __import__("hashlib").sha1(first.tobytes()).hexdigest()
        if r == 0:
            whole = fr.Context(m, device=dev)
            whole.fill_tables(fr.FILL_HASH, SEED_TABLES)
            whole.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            w0 = fr.Worker(whole, 512)
            ref = w0.infer(reqs[0][0][:512], reqs[0][1][:512])
            res["fp32_max_rel_err_vs_unsharded_first_512"] = float(np.abs(first[:512] - ref).max() / max(np.abs(ref).max(), 1e-30))
            w0.close()
            whole.close()
        ctx.set_fc_precision(fr.FC_BF16)
        for i in range(5):
            wk.infer_sharded(comm, *reqs[i % 4])
        steps = 40
        t0 = time.perf_counter()
        for i in range(steps):
            wk.idx[:B] = reqs[i % 4][0]
            wk.dense[:B] = reqs[i % 4][1]
            wk.submit_sharded(comm, B)
            wk.sync()
        dt = time.perf_counter() - t0
        res.update({"ok": True, "dtype": "bf16", "steps": steps, "ms_per_step": 1e3 * dt / steps, "inferences_per_s": B * steps / dt,
                    "what": "fr_worker_submit_sharded + fr_worker_sync per step through the C-ABI's own RCCL communicator (host-buffer form: the request's "
                            "H2D and the scores' D2H inside), Model-C batch 4096, %d table-ID shards, bf16 slice transport" % G})
        wk.close()
        comm.close()
        ctx.close()
    except BaseException as ex:   # noqa: BLE001
        res.update({"ok": False, "error": repr(ex)[:400]})
    with open(out_path + ".tmp", "w") as f:
        json.dump(res, f)
    os.replace(out_path + ".tmp", out_path)


def leg_cabi_rccl(fr, env, local_rank, limit_s=150.0):
    """N > 1, real devices: the C-ABI's own RCCL step (fr_comm_init_rank / fr_worker_submit_sharded) with G = N ranks, each rank in a CHILD process
    (cabi_child_main) so that nothing it does can cost the line.  Rank 0 draws the RCCL unique id and broadcasts it through the process group.
    -> rank 0: the child's result object (or why there is none); other ranks: None."""
    import tempfile
    idhex = [fr.Comm.unique_id().hex() if env.rank == 0 else None]
    env.dist.broadcast_object_list(idhex, src=0)
    out = os.path.join(tempfile.gettempdir(), "fr_cabi_%s_%d_%d.json" % (os.environ.get("MASTER_PORT", "0"), os.getpid(), env.rank))
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cabi-child", "%d/%d/%d/%s/%s" % (env.rank, env.world, local_rank, idhex[0], out)],
                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        _, err = child.communicate(timeout=limit_s)
        why = None if child.returncode == 0 else "child exited %s: %s" % (child.returncode, err.decode(errors="replace")[-300:])
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        why = "no result within %.0f s (child killed)" % limit_s
    res = None
    if os.path.exists(out):
        try:
            res = json.load(open(out))
        finally:
            os.unlink(out)
    if res is None:
        res = {"ok": False, "error": why or "the child wrote no result"}
    ok_all = env.sum_over_ranks(1.0 if res.get("ok") else 0.0)      # (the parents' own process group: every rank reports its child)
    res["ranks_ok"] = int(ok_all)
    return res if env.rank == 0 else None


def main_sharded(args, graft):
    """`--mode sharded`: the table-sharded configuration alone (see run_sharded); prints its own JSON line."""
    import importlib
    fr = graft.load_package()
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.force_process_group:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    env = dist_mod.DistEnv(args.backend if (world > 1 or args.force_process_group) else None, force_group=args.force_process_group)
    if args.device == -1:
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        fr.cpu_set_threads(max(1, usable // world))
        dev_id = fr.DEVICE_CPU
        args.precision = "f32"
    else:
        n_dev = max(fr.device_count(), 1)
        dev_id = env.local_rank % n_dev if args.share_device else env.local_rank
    if args.batch == 256:
        args.batch = 4096
    res = run_sharded(fr, dist_mod, env, dev_id, args)
    if env.rank == 0:
        print(json.dumps(res))
    env.close()


def run_sharded(fr, dist_mod, env, dev_id, args, auto_steps_s=0.0):
    """Model-C, batch 4096, tables sharded by table-ID over the ranks (BASELINE configs[3] / [4]; the reference's counterpart is the
    3-source ingest of GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/cuda_server.c:513-591); per step every rank gathers its
    [B x F] slice, ONE RCCL all-gather (or all-to-all) over xGMI delivers the slices, rank r runs the FC chain on its B/G items.
    value = B x steps / max-over-ranks time (one batch per step for the whole job: "scaling": "strong").
    args: batch, steps, warmup, precision, transport, exchange, backend, rows_cap, row_scale, no_unsharded_check.  auto_steps_s > 0
    sizes the timed region to that many seconds from the warm-up steps (the same count on every rank: MAX over ranks).
    Every rank calls this with the same arguments; the contexts are created and closed inside.  -> the result dict (every rank)."""
    import torch
    if dev_id == fr.DEVICE_CPU:
        return run_sharded_cpu(fr, dist_mod, env, args)
    world = env.world
    G, r = env.world, env.rank
    B = args.batch
    steps, warmup = max(args.steps, 1), args.warmup
    model = fr.Model.builtin(fr.MODEL_C)
    if args.rows_cap or args.row_scale != 1.0:
        # --row-scale 5: BASELINE configs[4] -- every table 5 x its rows: 316 GB in total, more than one GPU's 288 GB, 30-141 GB per 8-way shard
        model = model.clone(row_scale=args.row_scale, max_rows=args.rows_cap)
    if args.row_scale > 1.0:
        args.no_unsharded_check = True   # the whole model no longer fits one GPU: that is the point of the configuration
    ctx = fr.Context(model, device=dev_id, shard_rank=r, n_shards=G)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    ctx.set_chain_width(1)   # ONE worker runs FC chains on this rank (the other one gathers): its GEMM launches have the chip to themselves
    offs, lens, F = model.shard_plan(G)
    dev = torch.device("cuda", dev_id)
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(SEED_IDX)             # same request stream on every rank (replicated request)
    rows = model.rows()
    nbuf = 8
    idx_host = [uniform_idx(rng, rows, B) for _ in range(nbuf)]
    dense_host = [rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32) for _ in range(nbuf)]
    idxs = [fr.DeviceBuffer.from_numpy(ctx, a) for a in idx_host]
    dns = [fr.DeviceBuffer.from_numpy(ctx, a) for a in dense_host]
    # Two workers and two sets of exchange buffers: while the FC chain of batch i-1 runs on worker B's stream, worker A gathers
    # batch i and the exchange of batch i runs on torch's (RCCL) stream -- gather + exchange are hidden behind the MFMA work.
    wk = fr.Worker(ctx, B)       # gathers
    wk_fc = fr.Worker(ctx, B)    # FC chains
    a2a = args.exchange == "alltoall"
    if a2a and B % G:
        raise SystemExit("--exchange alltoall needs the batch divisible by the number of ranks")
    lp = args.transport == "lp" and args.precision != "f32"   # slices travel in the chain's own operand type
    prec_enum = {"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[args.precision]
    esz = {"f32": 4, "bf16": 2, "fp8": 1}[args.precision] if lp else 4
    local = [torch.empty((B, F * esz), dtype=torch.uint8, device=dev) for _ in range(2)]   # torch owns the exchange buffers (RCCL plumbing)
    gathered = [torch.empty((G, B // G if a2a else B, F * esz), dtype=torch.uint8, device=dev) for _ in range(2)]
    lo, hi = dist_mod.item_range(r, G, B)
    scores = torch.empty((max(hi - lo, 1),), dtype=torch.float32, device=dev)
    gloo = env.dist is not None and args.backend == "gloo"   # plumbing test: gloo moves host tensors

    def xchg(fn, src, dst):
        if gloo:
            torch.cuda.synchronize()
            h = fn(src.cpu())
            dst.copy_(h.view(dst.shape).to(dev))
        else:
            fn(src, dst)

    def exchange(k):
        xchg(env.all_to_all_slices if a2a else env.all_gather_slices, local[k], gathered[k])

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
    fc_events = [torch.cuda.Event(), torch.cuda.Event()]   # one per buffer set, re-recorded every other step (the host loop is what bounds small steps)

    def step(i):
        k = i & 1
        if state["fc_done"][k] is not None:
            s_gather.wait_event(state["fc_done"][k])   # the FC chain that read gathered[k] / local[k] two steps ago has finished
        wk.gather_slices(B, idxs[i % nbuf], dns[i % nbuf], local[k].data_ptr(), prec_enum if lp else fr.FC_FP32)   # async, gather worker's stream
        if state["pending"] is not None:
            p = state["pending"]
            s_fc.wait_stream(s_x)                      # its exchange has landed
            fc(p)                                      # async on the FC worker's stream: overlaps this step's gather + exchange
            fc_events[p].record(s_fc)
            state["fc_done"][p] = fc_events[p]
        s_x.wait_stream(s_gather)                      # the slice must be complete before RCCL reads it
        exchange(k)                                    # on s_x: torch's CURRENT stream for the whole loop (set below; the workers' launches name their own streams)
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

    if args.precision != "f32":
        ctx.set_fc_precision(prec_enum)
        if args.precision == "fp8":
            # activation exponents from the first batch's gathered fp32 slices; every rank must end up with the SAME exponents (a slice
            # encoded with one rank's X exponent is decoded with the receiver's): MIN over ranks = the most conservative scale
            cal_l = torch.empty((B, F), dtype=torch.float32, device=dev)
            cal_g = torch.empty((G, B // G if a2a else B, F), dtype=torch.float32, device=dev)
            wk.gather_only(B, idxs[0], dns[0], cal_l.data_ptr())
            wk.sync()
            xchg(env.all_to_all_slices if a2a else env.all_gather_slices, cal_l, cal_g)
            torch.cuda.synchronize()
            if a2a:
                wk_fc.calibrate_fp8_slices(B // G, 0, hi - lo, cal_g.data_ptr())
            else:
                wk_fc.calibrate_fp8_slices(B, 0, B, cal_g.data_ptr())
            act, _ = ctx.fp8_exponents()
            ctx.set_fp8_act_exponents(env.min_over_ranks_int(act))
            del cal_l, cal_g
    prev_stream = torch.cuda.current_stream(dev)
    torch.cuda.set_stream(s_x)   # the exchange stream is torch's current stream from here to the verification below (no per-step context manager)
    tw = time.perf_counter()
    for i in range(warmup):
        step(i)
    drain()
    if auto_steps_s > 0:   # size the timed region from the warm-up's rate; every rank must run the same number of steps
        per = (time.perf_counter() - tw) / max(warmup, 1)
        steps = int(env.max_over_ranks(float(min(max(int(auto_steps_s / max(per, 1e-6)), 5), 4000))))
    env.barrier(); torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    drain()                                            # the last batch's FC chain is inside the timed region
    env.barrier(); torch.cuda.synchronize(); ctx.synchronize()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    # outside the timed region: (1) the last pipelined batch against the same batch run step by step with host synchronisation;
    # (2) this rank's scores against an UNSHARDED context of the same model fed the same batch (rank 0 only: a second copy of the tables)
    piped = scores.clone()
    last = steps - 1
    wk.gather_slices(B, idxs[last % nbuf], dns[last % nbuf], local[0].data_ptr(), prec_enum if lp else fr.FC_FP32)
    wk.sync()
    exchange(0)
    torch.cuda.synchronize()
    fc(0)
    wk_fc.sync()
    torch.cuda.synchronize()
    verified = bool(torch.equal(piped, scores))
    torch.cuda.set_stream(prev_stream)
    vs_unsharded = None
    if r == 0 and not args.no_unsharded_check:
        full = fr.Context(model, device=dev_id)
        full.fill_tables(fr.FILL_HASH, SEED_TABLES)
        full.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
        if args.precision != "f32":
            full.set_fc_precision(prec_enum)
            if args.precision == "fp8":
                full.set_fp8_act_exponents(ctx.fp8_exponents()[0])
        fw = fr.Worker(full, B)
        ref = fw.infer(idx_host[last % nbuf], dense_host[last % nbuf])[lo:hi]
        got = scores.cpu().numpy()[:hi - lo]
        vs_unsharded = {"max_rel_err": float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)), "bit_identical": bool(np.array_equal(got, ref))}
        fw.close()
        full.close()
    res = {
        "metric": "inferences/sec, Model-C batch %d, tables sharded by table-ID" % B, "value": B * steps / dt, "unit": "inferences/s",
        "n_gpus": G, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "Model-C%s batch=%d, %d-way table-ID shards (slice F=%d floats), 1 %s of [B x F] per step, "
                               "FC on B/G items per rank" % (" (rows x %g: %.0f GB of tables)" % (args.row_scale, model.table_bytes() / 1e9) if args.row_scale != 1.0 else "",
                                                             B, G, F, "all-to-all" if a2a else "all-gather"), "parallelism": "table-sharded x%d" % G,
                   "shard_table_bytes_this_rank": int(sum(t.rows * t.dim * 4 for si in model.segments() if si.kind == fr.SEG_TABLE and offs[r] <= si.rec_offset < offs[r] + lens[r]
                                                        for t in [model.tables()[si.src]])),
                   "min_shards_for_288GB": model.min_shards(),   # north_star: shard ONLY when the tables outgrow one GPU (1 = replicas would do)
                   "exchange": args.exchange, "backend": args.backend if env.dist is not None else None,
                   "slice_transport": args.precision if lp else "f32", "pipelined_equals_stepwise": verified,
                   "sharded_vs_unsharded_context": vs_unsharded,
                   "exchange_bytes_in_per_rank_per_step": int(G * (B // G if a2a else B) * F * esz)}}
    wk.close()
    wk_fc.close()
    for b_ in idxs + dns:
        b_.free()
    ctx.close()
    del local, gathered, scores
    torch.cuda.empty_cache()
    return res


def run_sharded_cpu(fr, dist_mod, env, args):
    """run_sharded on the library's CPU back-end (--device -1: a REHEARSAL of the table-sharded step's control flow and plan arithmetic with
    any number of ranks and no GPU): per step every rank gathers its [B x F] fp32 slice (fr_worker_gather_slices on a CPU context), ONE
    all-gather (or all-to-all) through the process group delivers the slices, rank r runs the fp32 chain on its B / G items
    (fr_worker_fc_from_slices_lp).  Steps run one after the other (nothing to overlap on a host).  Same result keys as run_sharded; the
    figures are not measurements of anything."""
    import torch
    G, r = env.world, env.rank
    B = args.batch
    steps, warmup = max(args.steps, 1), args.warmup
    if args.precision != "f32":
        raise SystemExit("--device -1: the CPU back-end computes in fp32")
    model = fr.Model.builtin(fr.MODEL_C)
    if args.rows_cap or args.row_scale != 1.0:
        model = model.clone(row_scale=args.row_scale, max_rows=args.rows_cap)
    ctx = fr.Context(model, device=fr.DEVICE_CPU, shard_rank=r, n_shards=G)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    offs, lens, F = model.shard_plan(G)
    rng = np.random.default_rng(SEED_IDX)
    rows = model.rows()
    nbuf = 2
    idx_host = [uniform_idx(rng, rows, B) for _ in range(nbuf)]
    dense_host = [rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32) for _ in range(nbuf)]
    a2a = args.exchange == "alltoall"
    if a2a and B % G:
        raise SystemExit("--exchange alltoall needs the batch divisible by the number of ranks")
    wk = fr.Worker(ctx, B)
    lo, hi = dist_mod.item_range(r, G, B)
    local = np.zeros((B, F), np.float32)
    scores = np.zeros(max(hi - lo, 1), np.float32)

    def step(i):
        wk.gather_slices(B, idx_host[i % nbuf].ctypes.data, dense_host[i % nbuf].ctypes.data, local.ctypes.data, fr.FC_FP32)
        wk.sync()
        g = (env.all_to_all_slices if a2a else env.all_gather_slices)(torch.from_numpy(local)).contiguous().numpy()
        if a2a:
            wk.fc_from_slices_lp(B // G, 0, hi - lo, g.ctypes.data, fr.FC_FP32, scores.ctypes.data)
        else:
            wk.fc_from_slices_lp(B, lo, hi - lo, g.ctypes.data, fr.FC_FP32, scores.ctypes.data)
        wk.sync()

    for i in range(warmup):
        step(i)
    env.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    vs_unsharded = None
    if r == 0 and not args.no_unsharded_check and args.row_scale == 1.0:   # this rank's items through an unsharded CPU context of the same model
        last = steps - 1
        full = fr.Context(model, device=fr.DEVICE_CPU)
        full.fill_tables(fr.FILL_HASH, SEED_TABLES)
        full.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
        fw = fr.Worker(full, max(hi - lo, 1))
        ref = fw.infer(idx_host[last % nbuf][lo:hi], dense_host[last % nbuf][lo:hi])
        vs_unsharded = {"max_rel_err": float(np.abs(scores[:hi - lo] - ref).max() / max(np.abs(ref).max(), 1e-30)), "bit_identical": bool(np.array_equal(scores[:hi - lo], ref))}
        fw.close()
        full.close()
    res = {"metric": "inferences/sec, Model-C batch %d, tables sharded by table-ID" % B, "value": B * steps / dt, "unit": "inferences/s", "n_gpus": G, "steps": steps,
           "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic (REHEARSAL on the CPU back-end)",
           "config": {"workload": "REHEARSAL on the CPU back-end: Model-C%s batch=%d, %d-way table-ID shards (slice F=%d floats), 1 %s of [B x F] per step, FC on B/G items per rank"
                                  % (" (rows x %g)" % args.row_scale if args.row_scale != 1.0 else "", B, G, F, "all-to-all" if a2a else "all-gather"),
                      "parallelism": "table-sharded x%d" % G, "shard_table_bytes_this_rank": int(sum(model.shard_table_bytes(G)[r:r + 1])), "min_shards_for_288GB": model.min_shards(),
                      "exchange": args.exchange, "backend": args.backend if G > 1 else None, "slice_transport": "f32", "pipelined_equals_stepwise": None,   # no pipelined form on the CPU rehearsal: nothing was compared
                      "sharded_vs_unsharded_context": vs_unsharded, "exchange_bytes_in_per_rank_per_step": int(G * (B // G if a2a else B) * F * 4),
                      "slice_offsets": offs, "slice_lens": lens, "items_this_rank": [lo, hi]}}
    wk.close()
    ctx.close()
    return res


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=2, help="host driver threads (reference THREAD_NUM = 4, constant.h:42)")
    ap.add_argument("--depth", type=int, default=2, help="workers (streams) each driver thread keeps in flight")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="replicas: BASELINE configs[1] (default, the headline metric); sharded: Model-C batch 4096 with tables "
                         "sharded by table-ID over the ranks + one RCCL all-gather of the looked-up slices (BASELINE configs[3])")
    ap.add_argument("--model", choices=["A", "B", "C"], default="A", help="A = BASELINE configs[1] (default headline); B/C: one other config, throughput + roofline leg only")
    ap.add_argument("--precision", choices=["f32", "bf16", "fp8"], default="f32",
                    help="FC chain arithmetic (bf16 = BASELINE configs[2], fp8 = configs[4]: e4m3, calibrated on the first batch)")
    ap.add_argument("--legs", default="all",
                    help="comma list of the extra legs rank 0 runs at N = 1: roofline,groups,pcie,tcp,cpu,configs,gather,bank (default all; 'none' = headline only)")
    ap.add_argument("--gather-law", choices=["all", "uniform", "zipf"], default="all", help="gather leg: per-table index law(s) to run (PMC passes: one law per kernel name)")
    ap.add_argument("--transport", choices=["f32", "lp"], default="lp",
                    help="sharded mode: slices travel as fp32, or (lp) in the chain's own operand type when --precision is bf16 / fp8")
    ap.add_argument("--exchange", choices=["allgather", "alltoall"], default="allgather",
                    help="sharded mode: all-gather every slice to every rank (BASELINE configs[3]) or all-to-all only each rank's items")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true", help="plumbing test: ranks share the visible GPU(s) (use with --backend gloo)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="--gpus 1: build the torch.distributed process group (--backend nccl = RCCL) for the ONE rank and run the N > 1 legs through it -- "
                         "a one-GPU rehearsal of the RCCL code path (device tensors, all_gather_into_tensor on RCCL's stream, the external-stream hand-over)")
    ap.add_argument("--device", type=int, default=None,
                    help="-1: REHEARSAL on the library's CPU back-end (fr_ctx_create device = -1), no GPU touched -- the whole control flow of the line (every "
                         "leg's collectives, votes, plan arithmetic, LineGuard) with fp32 CPU contexts, row-capped tables and token step counts; use with "
                         "--backend gloo --rows-cap N.  The numbers of such a line are not measurements of anything")
    ap.add_argument("--rows-cap", type=int, default=0, help="plumbing tests: cap every table's row count (sharded mode)")
    ap.add_argument("--row-scale", type=float, default=1.0,
                    help="sharded mode: multiply every table's row count (BASELINE configs[4]: 5 inflates Model-C to 316 GB, past one GPU's 288 GB)")
    ap.add_argument("--no-unsharded-check", action="store_true", help="sharded mode: skip rank 0's comparison against an unsharded context")
    ap.add_argument("--no-gather-ab", action="store_true", help="gather legs: skip the kernel A/B (PMC passes: one kernel per leg)")
    ap.add_argument("--group", type=int, default=0, help="batches per fused launch (fr_ctx_set_stream_group); 0 = the context's default")
    ap.add_argument("--roofline-only", action="store_true",
                    help="rocprofv3 --kernel-trace --stats runs: nothing but the single-stream roofline launches touches the kernels being priced (the "
                         "multi-stream throughput loops, whose concurrent launches stretch each other, are skipped), so the profiler's average agrees with "
                         "the HIP-event figure on the bench line")
    ap.add_argument("--throughput-only", action="store_true", help="single-configuration runs: the multi-stream throughput loop alone, no one-stream roofline leg (profiled: *_4streams_kernel_stats.csv)")
    ap.add_argument("--quick", action="store_true", help="profiling runs: 0.3 s instead of >= 2 s behind `value` (the legs are what is being profiled)")
    ap.add_argument("--per-bank", action="store_true", help="single-configuration runs (--model / --precision): FR_INDEX_PER_BANK context and indices")
    ap.add_argument("--no-multi-gather", action="store_true", help="N > 1: skip the per-rank legs (gather_per_bank_all_ranks, configs_all_ranks)")
    ap.add_argument("--no-multi-configs", action="store_true", help="N > 1: skip configs_all_ranks (Model-B bf16, Model-C bf16 / fp8 on every rank)")
    ap.add_argument("--no-multi-sharded", action="store_true", help="N > 1: skip the table-sharded legs (`sharded`, `sharded_inflated_fp8`) of the default line")
    ap.add_argument("--fail-rank", type=int, default=-1, help="--plumbing-only: this rank exits with status 3 before the first barrier (launcher fail-fast test)")
    ap.add_argument("--fail-sharded-rank", type=int, default=-1, help="--plumbing-only: this rank raises inside a guarded collective leg (line + non-zero exit status test)")
    ap.add_argument("--cabi-child", default=None, help=argparse.SUPPRESS)  # RANK/WORLD/DEVICE/IDHEX/OUT: one rank of the C-ABI RCCL leg (cabi_child_main)
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)   # K/P/BUDGET: one process of the all-cores CPU baseline (cpu_child_main)
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch / rendezvous / timing-rule check without touching a GPU or the library (CPU test of the multi-GPU launcher)")
    args = ap.parse_args()
    if args.cpu_child:   # a host-only helper process of leg_cpu_baseline: before anything could touch a GPU
        return cpu_child_main(args.cpu_child)
    if args.cabi_child:  # one rank of the C-ABI RCCL leg, in a process of its own
        return cabi_child_main(args.cabi_child)

    # ---- rank processes: torchrun supplies WORLD_SIZE; otherwise start them ourselves, before anything touches the GPU ----
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report a different n_gpus than requested" % (args.gpus, world_env))

    import importlib
    import __graft_entry__ as graft
    if args.plumbing_only:
        return main_plumbing(args, graft)
    if args.mode == "sharded":
        return main_sharded(args, graft)
    fr = graft.load_package()
    dist_mod = importlib.import_module("fleetrec_amd.dist")
    if args.force_process_group:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    env = dist_mod.DistEnv(args.backend if (world_env > 1 or args.force_process_group) else None, force_group=args.force_process_group)
    rank, world = env.rank, env.world
    multi = world > 1 or args.force_process_group   # the N > 1 legs (one rank with --force-process-group: the RCCL code path rehearsed on one GPU)
    cpu = args.device == -1   # rehearsal on the CPU back-end: see --device
    if cpu:
        if args.model != "A" or args.precision != "f32":
            raise SystemExit("--device -1 rehearses the default line (the CPU back-end computes in fp32)")
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        fr.cpu_set_threads(max(1, usable // world))
        local_rank = fr.DEVICE_CPU
        if args.legs == "all":
            args.legs = "none"
    else:
        if fr.device_count() < 1:
            raise SystemExit("bench.py needs an MI355X: a context on device >= 0 never falls back to the CPU (--device -1 rehearses the line on the CPU back-end)")
        n_dev = fr.device_count()
        local_rank = env.local_rank % n_dev if args.share_device else env.local_rank  # --share-device: plumbing test on one GPU
    legs = set() if args.legs == "none" else set(args.legs.split(","))
    leg_s, t_leg = {}, [time.perf_counter()]   # wall time per leg (rank 0's clock): the budget the driver's timeout is held against

    def leg_done(name):
        now = time.perf_counter()
        leg_s[name] = round(leg_s.get(name, 0.0) + now - t_leg[0], 2)
        t_leg[0] = now
    # N > 1: the line carries the `roofline` object as well -- rank 0's HIP-event leg of the dominant kernel on its own replica while the
    # other ranks wait at the next barrier (one rank's launches on one stream: the same measurement as at N = 1); the other rank-0 legs are N = 1 only
    want = lambda name: rank == 0 and ("all" in legs or name in legs) and (not multi or name == "roofline")

    B = args.batch
    which = {"A": fr.MODEL_A, "B": fr.MODEL_B, "C": fr.MODEL_C}[args.model]
    model = fr.Model.builtin(which)
    if cpu and args.rows_cap:
        model = model.clone(max_rows=args.rows_cap)
    if args.per_bank:   # the reference kernel's index contract: one index per memory bank per item, bank-interleaved tables
        if args.model == "A" and args.precision == "f32":
            raise SystemExit("--per-bank: with --model B/C or a --precision other than f32 (the headline's 47 tables are 47 banks)")
        model = model.clone(index_mode=fr.INDEX_PER_BANK)
    ctx = fr.Context(model, device=local_rank)
    ctx.fill_tables(fr.FILL_HASH, SEED_TABLES)
    ctx.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
    rng = np.random.default_rng(dist_mod.replica_seed(SEED_IDX, rank))
    rows = model.bank_map()[1] if args.per_bank else model.rows()
    # rotating index buffers: enough that a launch group does not re-hit its own rows in the 256 MB Infinity Cache (Model-B: 16 buffers x 1024
    # items x 98 rows = 205 MB of lines FIT there, and the single-configuration runs read 353 M inf/s / 390 us per launch where the default
    # line's 32 buffers give 330 M / 411 us -- round 6 found the profiled runs flattered by that; SURVEY 8(d): ">= 32 rotating")
    n_bufs = N_IDX_BUFFERS if args.model == "A" else (32 if args.model == "B" else 16)
    idx_host = [uniform_idx(rng, rows, B) for _ in range(n_bufs)]
    d_idx = [fr.DeviceBuffer.from_numpy(ctx, a) for a in idx_host]
    dense_host = [rng.uniform(-1, 1, (B, model.dense_len)).astype(np.float32) for _ in range(n_bufs)] if model.dense_len else None
    d_dense = [fr.DeviceBuffer.from_numpy(ctx, a) for a in dense_host] if dense_host else None

    def barrier():
        env.barrier()
        ctx.synchronize()

    if args.group > 0:
        ctx.set_stream_group(args.group)
    elif args.precision == "bf16" and args.model in ("A", "B"):
        ctx.set_stream_group(bf16_launch_group(B))   # the group the default line's bf16 rows run at (so that profiled single-configuration runs match them)
    if args.model != "A" or args.precision != "f32":
        # one non-headline configuration on its own: throughput + its roofline leg
        res = leg_config(fr, ctx, model, B, args.precision, d_idx, d_dense, idx_host[0], dense_host[0] if dense_host else None, args.threads, args.depth,
                         "Model-%s batch=%d %s FC chain, %s, index rows resident in HBM, %d batches per launch" % (args.model, B, args.precision, "one index per bank" if args.per_bank else "per-table indices", ctx.stream_group()),
                         min_s=0.0 if args.roofline_only else (0.05 if args.quick else 1.0), env=env, roofline=not args.throughput_only)
        if rank == 0:
            print(json.dumps({"metric": "inferences/sec", "value": res["value"], "unit": "inferences/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": args.precision, "data": "synthetic", "config": {"workload": res["workload"]}, "timed_batches": res["timed_batches"],
                              "timed_s": res["timed_s"], "roofline": res.get("roofline"), "fc_tflops_end_to_end": res["fc_tflops_end_to_end"],
                              "layer_launch_ms": res.get("layer_launch_ms"), "layer_kernels": res.get("layer_kernels"), "layer_concurrency": res.get("layer_concurrency"),
                              "layer_stream_busy_ms": res.get("layer_stream_busy_ms")}))
        ctx.close()
        env.close()
        return

    # ---- headline: Model-A batch 256 fp32, index rows resident in HBM, native driver loop ---------------------------------------
    if args.roofline_only:   # one stream only, a token number of batches: see --roofline-only
        args.threads, args.depth, args.steps, args.warmup, args.quick = 1, 1, 64, 64, True
        if args.legs == "all":
            args.legs = "roofline"
            legs.clear()
            legs.add("roofline")
    driver = fr.Driver(ctx, args.threads, args.depth, B)
    driver.run_resident(B, max(args.warmup, 0), d_idx)
    barrier()
    t0 = time.perf_counter()
    driver.run_resident(B, max(args.steps, 1), d_idx)            # exactly --steps batches: the burst figure
    barrier()
    burst_dt = env.max_over_ranks(time.perf_counter() - t0)
    # (the calibration run inside steady_run includes the ramp and overestimates the time per batch by ~8 %: size for 1.12 x the target)
    n_timed = args.steps if (args.roofline_only or cpu) else max(steady_run(lambda k: driver.run_resident(B, k, d_idx), 0.3 if args.quick else 1.12 * STEADY_S, n_first=8192, env=env),
                                                        args.steps)
    barrier()
    t0 = time.perf_counter()
    driver.run_resident(B, n_timed, d_idx)                       # >= 2 s of back-to-back batches: `value`
    barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)

    leg_done("headline")
    result = None
    if rank == 0:
        result = {
            "metric": "inferences/sec at batch 256", "value": world * n_timed * B / dt, "unit": "inferences/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / n_timed, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Model-A (embedding_47_krnl: 47 tables, 1.415 GB) batch=%d, fp32 FC 352-1024-512-256-1, all tables resident in "
                                   "one GPU's HBM; hash-filled tables, uniform per-table indices; index rows resident in HBM when the timed region "
                                   "starts (%d rotating buffers), scores left in HBM (the bench contract's definition of `value`); value_pcie_inclusive "
                                   "= the reference's own loop (cuda_server.c:460-461,494-495): the same stream fed from host memory with scores "
                                   "delivered to host memory" % (B, n_bufs),
                       "batch": B, "driver_threads": args.threads, "workers_per_thread": args.depth,
                       "batches_per_launch": ctx.stream_group(), "parallelism": "replicas x%d" % world},
            "timed_batches": n_timed, "timed_s": dt, "leg_seconds": leg_s,
            "value_is": "steady state: %d back-to-back batches per rank over %.2f s (sized for >= %.1f s from a calibration run, whatever --steps says)" % (n_timed, dt, STEADY_S),
            "burst": {"steps": args.steps, "warmup": args.warmup, "value": world * args.steps * B / burst_dt, "unit": "inferences/s",
                      "ms_per_step": 1e3 * burst_dt / max(args.steps, 1),
                      "what": "EXACTLY --steps batches after --warmup batches, barrier + device sync on both sides: with few steps this is launch "
                              "latency (a fused launch carries %d batches), not throughput" % ctx.stream_group()},
        }

    # (the host-fed twin of `value` is measured right behind it, before the other legs have created and destroyed their streams)
    if want("pcie"):
        # ---- PCIe-inclusive rates (index rows start in HOST memory, scores end in HOST memory; reported, never `value`) ----
        # host threads of these legs = the reference's THREAD_NUM = 4 (constant.h:42): every pushed batch is first copied into pinned
        # staging by its driver thread (the counterpart of the reference's read() into pinned memory), and two threads cannot stage
        # 13 GB/s of index rows (measured: 2 x 2 workers 60.7 M inf/s, 4 x 2 workers 67.1 M, profiles/archive/r02_experiments.md section 5)
        ht = max(args.threads, 4)
        hd = fr.Driver(ctx, ht, 4, B)
        hd.run_host(B, 200, idx_host)
        n = steady_run(lambda k: hd.run_host(B, k, idx_host), 1.0, n_first=1000, quantum=64)
        el = hd.run_host(B, n, idx_host)
        hd.close()
        result["pcie_inclusive"] = {"value": n * B / el, "unit": "inferences/s", "ms_per_step": 1e3 * el / n, "timed_batches": n, "timed_s": el,
                                    "what": "per batch: memcpy to pinned -> fr_worker_submit (5 stage launches; index rows read from and scores written to the pinned "
                                            "buffers over PCIe) -> sync (the reference's own per-batch sequence, cuda_server.c:425-495), %d threads x 4 workers" % ht}
        # the reference's loop with its two PCIe hops inside (cuda_server.c:460-461,494-495), streamed: 4 host threads (the reference's THREAD_NUM)
        # x ONE worker each = one stream per hardware queue -- a block's H2D runs on the worker's copy stream ahead of its launch, the scores
        # are written straight to pinned memory, the worker's stream carries nothing but kernels (profiles/r05_host_fed_timeline.txt:
        # 99.6-100 % of the HBM-resident rate; round 4's three commands per block on 4 x 2 streams: 94-96 %)
        hs = fr.Driver(ctx, ht, 1, B)
        hs.run_host(B, 2048, idx_host, streaming=True)
        n = steady_run(lambda k: hs.run_host(B, k, idx_host, streaming=True), STEADY_S, n_first=8192)
        el = hs.run_host(B, n, idx_host, streaming=True)
        hs.close()
        result["value_pcie_inclusive"] = n * B / el   # the same metric with the reference loop's H2D / D2H inside (never `value`: bench contract)
        result["value_hbm_resident"] = result["value"]
        result["pcie_inclusive_streaming"] = {"value": n * B / el, "unit": "inferences/s", "ms_per_step": 1e3 * el / n, "timed_batches": n, "timed_s": el,
                                              "fraction_of_value": n * B / el / result["value"],
                                              "what": "host-resident request stream, scores delivered to host memory: blocks of 64 batches staged in pinned "
                                                      "memory, one H2D (copy stream) + one fused launch per block, scores written to pinned memory by the kernel "
                                                      "(fr_worker_push_host), %d threads x 1 worker" % ht}

    leg_done("pcie")
    if rank == 0 and cpu:
        result["data"] = "synthetic (REHEARSAL on the CPU back-end: control flow only, no figure of this line is a measurement)"
        result["config"]["rehearsal"] = "fr_ctx_create(device = -1) on every rank, fp32, rows capped at %d, token step counts" % args.rows_cap
        result["roofline"] = {"bound": "none", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None}
        result["cpu_baseline"] = {"value": None, "unit": "inferences/s", "cores": 0, "kind": "port", "sample": "none (rehearsal)"}
    gpu_scores_first = None
    if want("roofline") and not cpu:
        # ---- roofline of the dominant kernel: measured live with HIP events on the worker's stream ----
        # Model-A streams through the fused item-tile kernel: ONE launch = the whole hot path (gather + 4 GEMMs) of `group` queued
        # batches, 64 items per workgroup at the default group of 64.
        group = ctx.stream_group()
        wk = fr.Worker(ctx, B)
        ring = [fr.DeviceBuffer(ctx, B * 4) for _ in range(max(8, 2 * group))]
        push = lambda i: wk.push_device(B, d_idx[i % n_bufs], None, ring[i % len(ring)])
        t_warm = time.time()   # >= 1 s of back-to-back launches first: the shader clock ramps over many milliseconds of load
        while time.time() - t_warm < 1.0:
            for i in range(4 * group):
                push(i)
            wk.sync()
        pipe_ms = time_launches(wk, push, group, 200, warm_launches=4, push_launch=launch_pusher(wk, B, d_idx[:n_bufs], None, ring, group))
        flops = fc_flops_per_inference(model.fc) * B * group
        ach = flops / (pipe_ms * 1e-3) / 1e12
        kname = wk.last_kernel()   # the kernel that carried these launches, as the library reports it
        pm = pmc("fused_m2_A256") or {}
        result["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TF["f32"], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TF["f32"],
                              "traffic": pm.get("traffic_bytes_per_launch"),
                              "traffic_source": "%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --legs roofline`; "
                                                "FETCH_SIZE x2 gfx950 correction), bytes per launch; read from the committed file, not measured in this run" % pm.get("pmc_file"),
                              "kernel": "%s: one launch = the whole hot path (gather + the 4-GEMM chain) of %d queued batches of %d, "
                                        "back-to-back on ONE stream" % (kname, group, B),
                              "kernel_name": kname, "profiled_avg_launch_us": profiled_avg_us(find_profile("roofline_kernel_stats.csv"), kname),
                              "profile": "profiles/" + find_profile("roofline_kernel_stats.csv"),
                              "profile_what": "rocprofv3 --kernel-trace --stats of `bench.py --roofline-only`; *_value_4streams_kernel_stats.csv is the same kernel "
                                              "under the DEFAULT four-stream `value` run",
                              "profiled_avg_launch_us_4streams": profiled_avg_us(find_profile("value_4streams_kernel_stats.csv"), kname),
                              "batches_per_launch": group, "avg_launch_ms": pipe_ms, "algorithmic_flops_per_launch": flops, "achieved_is": "one_launch_alone_on_one_stream",
                              "pmc_mfma_busy_fraction": pm.get("mfma_busy_fraction"), "pmc_mfma_f32_flops_per_launch": pm.get("mfma_f32_flops_per_launch"),
                              "note": "`value` above runs %d such streams concurrently" % (args.threads * args.depth)}
        wk.sync()
        # scores of the first index buffer, for the CPU leg's cross-check
        d_sc = fr.DeviceBuffer(ctx, B * 4)
        wk.submit_device(B, d_idx[0], None, d_sc)
        wk.sync()
        gpu_scores_first = d_sc.download(np.float32, B)
        d_sc.free()
        wk.close()
        for b_ in ring:
            b_.free()

    leg_done("roofline")
    if want("groups"):
        result["launch_group_table"] = {"workload": "Model-A batch 256 fp32, %d threads x %d workers; group = fr_ctx_set_stream_group: below 12 every push is one "
                                                    "stage-pipeline launch, from 12 up a group is one fused launch; launch_ms_one_stream = time per group of pushes"
                                                    % (args.threads, args.depth),
                                        "rows": leg_group_table(fr, ctx, model, B, d_idx, args.threads, args.depth)}
        for row in result["launch_group_table"]["rows"]:   # latency price of the launch group `value` runs at: first push -> all scores, idle worker
            if row["group"] == result["config"]["batches_per_launch"]:
                result["config"]["push_to_scores_us_p50"] = 1e3 * row["push_to_scores_ms_p50"]

    leg_done("groups")
    if want("tcp"):
        try:
            result["tcp_streaming"] = leg_tcp(B, local_rank)
        except Exception as ex:
            result["tcp_streaming"] = {"error": repr(ex)[:300]}
        try:
            result["tcp_serving_with_replies"] = leg_tcp_reply(B, local_rank)
        except Exception as ex:
            result["tcp_serving_with_replies"] = {"error": repr(ex)[:300]}

    leg_done("tcp")
    if want("cpu"):
        try:
            result["cpu_baseline"] = leg_cpu_baseline(graft, ctx, model, idx_host, B, gpu_scores_first)
        except Exception as ex:
            result["cpu_baseline"] = {"error": repr(ex)}
        try:   # BASELINE configs[0] and the same sample through the library's own CPU back-end (device = -1): product code, not the checker
            result["cpu_backend"] = leg_cpu_backend(fr, idx_host, B, gpu_scores_first)
            if isinstance(result.get("cpu_baseline"), dict):
                result["cpu_baseline"]["batch_1"] = result["cpu_backend"]["batch_1"]
        except Exception as ex:
            result["cpu_backend"] = {"error": repr(ex)[:300]}
        try:   # the same single request on the GPU path: fr_worker_submit + sync of a batch of 1 (index row H2D, 5 stage launches, score D2H)
            w1 = fr.Worker(ctx, 1)
            one = idx_host[0][:1]
            for _ in range(50):
                w1.infer(one)
            ts = []
            for _ in range(300):
                t1 = time.perf_counter()
                w1.infer(one)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            w1.close()
            result["batch_1_gpu"] = {"us_p50": 1e6 * ts[len(ts) // 2], "us_p90": 1e6 * ts[9 * len(ts) // 10],
                                     "what": "Model-A batch 1 on the GPU path: fr_worker_submit + fr_worker_sync from the host (PCIe hops and the ctypes call included)"}
        except Exception as ex:
            result["batch_1_gpu"] = {"error": repr(ex)[:200]}

    leg_done("cpu")
    driver.close()
    for b in d_idx:
        b.free()
    ctx.close()

    # ---- the other BASELINE configurations, each with its own in-run roofline object (rank 0, N = 1 only) ----------------------------
    if want("configs"):
        cfgs = []
        try:
            mb = fr.Model.builtin(fr.MODEL_B)
            cb = fr.Context(mb, device=local_rank)
            cb.fill_tables(fr.FILL_HASH, SEED_TABLES)
            cb.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            rngb = np.random.default_rng(SEED_IDX)
            ih = [uniform_idx(rngb, mb.rows(), 1024) for _ in range(32)]
            di = [fr.DeviceBuffer.from_numpy(cb, a) for a in ih]
            for prec in ("bf16", "f32", "fp8"):
                cb.set_stream_group(bf16_launch_group(1024) if prec != "f32" else 64)   # (fp8: the chunked kernel, whose launches carry 16 batches of 1024 each: the group only deepens the queue)
                cfgs.append(leg_config(fr, cb, mb, 1024, prec, di, None, ih[0], None, args.threads, args.depth,
                                       "BASELINE configs[2]: Model-B (embedding_98_krnl, 15.1 GB) batch=1024, %s FC, fused concat + FC chain, per-table indices, %d batches per launch (fr_ctx_set_stream_group)" % (prec, bf16_launch_group(1024))
                                       if prec == "bf16" else "Model-B batch=1024, %s FC%s, per-table indices" % (prec, " (the reference's own precision)" if prec == "f32" else " through the chunked fused kernel, launch group %d" % bf16_launch_group(1024)),
                                       pmc_key="fused_h_B1024_bf16" if prec == "bf16" else None, profile_csv=find_profile("B1024_%s_kernel_stats.csv" % prec), tag="B1024_" + prec))
            cb.close()
            # the same configuration under the reference kernel's index contract: one index per bank (49 banks of 2 tables), bank rows in HBM
            mbb = mb.clone(index_mode=fr.INDEX_PER_BANK)
            cb = fr.Context(mbb, device=local_rank)
            cb.fill_tables(fr.FILL_HASH, SEED_TABLES)
            cb.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            ihb = [uniform_idx(rngb, mbb.index_ranges(), 1024) for _ in range(32)]
            dib = [fr.DeviceBuffer.from_numpy(cb, a) for a in ihb]
            cb.set_stream_group(bf16_launch_group(1024))
            cfgs.append(leg_config(fr, cb, mbb, 1024, "bf16", dib, None, ihb[0], None, args.threads, args.depth,
                                   "BASELINE configs[2] under the kernel's index contract: Model-B batch=1024, bf16 FC, ONE index per bank (FR_INDEX_PER_BANK, 49 banks), %d batches per launch" % bf16_launch_group(1024), tag="B1024_bf16_per_bank", profile_csv=find_profile("B1024_bf16_per_bank_kernel_stats.csv")))
            cb.close()
        except Exception as ex:
            cfgs.append({"workload": "Model-B", "error": repr(ex)})
        try:   # the headline workload in the low-precision chains (reported beside `value`, never as it: BASELINE configs[1] says fp32 FC)
            ma = fr.Model.builtin(fr.MODEL_A)
            ca = fr.Context(ma, device=local_rank)
            ca.fill_tables(fr.FILL_HASH, SEED_TABLES)
            ca.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            rnga = np.random.default_rng(SEED_IDX)
            iha = [uniform_idx(rnga, ma.rows(), B) for _ in range(N_IDX_BUFFERS)]
            dia = [fr.DeviceBuffer.from_numpy(ca, a_) for a_ in iha]
            for prec in ("bf16", "fp8"):
                # bf16: a launch group of 256 batches = four 64-item tiles per compute unit per launch, which is what the persistent kernel
                # (fr_fused_tile_hs_kernel) needs to overlap one tile's gather with another's FC phases; at the default 64 (one tile per compute
                # unit, the chunked kernel) the same leg gives 434-436 M inf/s against 489 M (profiles/archive/r03_group_above_64_ab.txt)
                ga = bf16_launch_group(B) if prec == "bf16" else 64
                ca.set_stream_group(ga)
                cfgs.append(leg_config(fr, ca, ma, B, prec, dia, None, iha[0], None, args.threads, args.depth,
                                       "Model-A batch=%d (the headline workload), %s FC chain through the fused item-tile kernel, %d batches per launch (fr_ctx_set_stream_group)" % (B, prec, ga), tag="A%d_%s" % (B, prec), profile_csv=find_profile("A%d_%s_kernel_stats.csv" % (B, prec))))
            ca.close()
        except Exception as ex:
            cfgs.append({"workload": "Model-A low precision", "error": repr(ex)})
        result["configs"] = cfgs
    leg_done("configs_A_B")

    if want("gather") or want("configs") or want("bank"):
        try:
            BC = 4096
            mc = fr.Model.builtin(fr.MODEL_C)
            if want("gather") or want("configs"):
                cc = fr.Context(mc, device=local_rank)
                cc.fill_tables(fr.FILL_HASH, SEED_TABLES)
                cc.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            if want("gather"):
                g = {"workload": "Model-C (2x embedding_377_krnl + 64 dense: 376 tables, 63.2 GB) batch=4096, record-producing gather "
                                 "(fr_worker_gather_only), per-table indices", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
                if args.gather_law in ("all", "uniform"):
                    g.update(leg_gather(fr, cc, mc, BC, "uniform", variants=not args.no_gather_ab))
                    pm = pmc("gather_C4096_per_table_uniform") or {}
                    g["traffic"] = pm.get("traffic_bytes_per_launch")
                    g["l2_hit_rate"] = pm.get("l2_hit_rate")
                    g["traffic_source"] = "%s (PMC passes of `bench.py --legs gather --gather-law uniform`, FETCH_SIZE x2 correction), bytes per launch" % pm.get("pmc_file")
                    g["profiled_avg_launch_us"] = profiled_avg_us(find_profile("gather_per_table_uniform_kernel_stats.csv"), g.get("kernel_name"))
                    g["profile"] = "profiles/" + find_profile("gather_per_table_uniform_kernel_stats.csv")
                if args.gather_law in ("all", "zipf"):
                    z = leg_gather(fr, cc, mc, BC, "zipf", seed=SEED_IDX + 1, variants=not args.no_gather_ab)
                    pm = pmc("gather_C4096_per_table_zipf") or {}
                    z["traffic"] = pm.get("traffic_bytes_per_launch")
                    z["l2_hit_rate"] = pm.get("l2_hit_rate")
                    z["profiled_avg_launch_us"] = profiled_avg_us(find_profile("gather_per_table_zipf_kernel_stats.csv"), z.get("kernel_name"))
                    z["profile"] = "profiles/" + find_profile("gather_per_table_zipf_kernel_stats.csv")
                    g["zipf_1.05"] = z
                result["gather"] = g
            if want("configs"):
                rngc = np.random.default_rng(SEED_IDX)
                ih = [uniform_idx(rngc, mc.rows(), BC) for _ in range(8)]
                dh = [rngc.uniform(-1, 1, (BC, mc.dense_len)).astype(np.float32) for _ in range(8)]
                di = [fr.DeviceBuffer.from_numpy(cc, a) for a in ih]
                dd = [fr.DeviceBuffer.from_numpy(cc, a) for a in dh]
                for prec in ("f32", "bf16", "fp8"):
                    result["configs"].append(leg_config(fr, cc, mc, BC, prec, di, dd, ih[0], dh[0], args.threads, args.depth,
                                                        "Model-C (63.2 GB, unsharded replica) batch=4096, %s FC chain end to end "
                                                        "(BASELINE configs[3]/[4] shapes on one GPU)" % prec, pmc_key="gemm_C4096_%s" % prec, profile_csv=find_profile("C4096_%s_kernel_stats.csv" % prec), tag="C4096_" + prec))
            if want("gather") or want("configs"):
                cc.close()
            if want("bank"):
                # the reference kernel's real index contract: ONE index per bank per item, bank-interleaved table layout
                mcb = mc.clone(index_mode=fr.INDEX_PER_BANK)
                cbk = fr.Context(mcb, device=local_rank)
                cbk.fill_tables(fr.FILL_HASH, SEED_TABLES)
                gb = {"workload": "Model-C batch=4096, FR_INDEX_PER_BANK: one index per memory bank per item (82 banks; embedding_377_krnl.cpp:1261-1290), "
                                  "tables of a bank row-interleaved in HBM; uniform indices over each bank's valid range", "bound": "hbm",
                      "peak": HBM_PEAK_GBS, "unit": "GB/s"}
                gb.update(leg_gather(fr, cbk, mcb, BC, "uniform", variants=not args.no_gather_ab))
                pm = pmc("gather_C4096_per_bank_uniform") or {}
                gb["traffic"] = pm.get("traffic_bytes_per_launch")
                gb["l2_hit_rate"] = pm.get("l2_hit_rate")
                gb["profiled_avg_launch_us"] = profiled_avg_us(find_profile("gather_per_bank_uniform_kernel_stats.csv"), gb.get("kernel_name"))
                gb["profile"] = "profiles/" + find_profile("gather_per_bank_uniform_kernel_stats.csv")
                result["gather_per_bank"] = gb
                if want("configs"):   # Model-C end to end under the bank contract (82 bank fetches per item instead of 376 rows)
                    cbk.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
                    rngk = np.random.default_rng(SEED_IDX)
                    ihk = [uniform_idx(rngk, mcb.index_ranges(), BC) for _ in range(8)]
                    dhk = [rngk.uniform(-1, 1, (BC, mcb.dense_len)).astype(np.float32) for _ in range(8)]
                    dik = [fr.DeviceBuffer.from_numpy(cbk, a) for a in ihk]
                    ddk = [fr.DeviceBuffer.from_numpy(cbk, a) for a in dhk]
                    for prec in ("bf16", "fp8"):
                        leg = leg_config(fr, cbk, mcb, BC, prec, dik, ddk, ihk[0], dhk[0], args.threads, args.depth,
                                         "Model-C batch=4096, %s FC chain end to end, ONE index per bank (FR_INDEX_PER_BANK, 82 banks); the in-chain gather reads the "
                                         "operand-type bank image (bank rows already %s: 82 lines per item instead of 142)" % (prec, "bf16" if prec == "bf16" else "e4m3"),
                                         tag="C4096_%s_per_bank" % prec)
                        # round 6: what the image costs in HBM, and the committed kernel-level A/B of the chain's gather launch (image on / off)
                        leg["lp_bank_image"] = {"bytes": cbk.lp_bank_image_bytes(), "fp32_table_bytes": int(mcb.table_bytes()),
                                                "gather_kernel_us_image_on": profiled_avg_us("r06_lp_image_%s_img1_lone_kernel_stats.csv" % prec, "fr_pipeline_kernel<0, %d>" % (1 if prec == "bf16" else 2)),
                                                "gather_kernel_us_image_off": profiled_avg_us("r06_lp_image_%s_img0_lone_kernel_stats.csv" % prec, "fr_pipeline_kernel<0, %d>" % (1 if prec == "bf16" else 2)),
                                                "profiles": "profiles/r06_lp_image_%s_img{0,1}_{lone,chains}_kernel_stats.csv" % prec}
                        result["configs"].append(leg)
                cbk.close()
        except Exception as ex:  # the main metric must still be reported
            result.setdefault("gather", {})["error"] = repr(ex)

    leg_done("gather_and_configs_C")
    if multi and not args.no_multi_gather:
        # N > 1: the other half of BASELINE.json's metric ("embedding-gather HBM GB/s vs peak, 1 -> 8 MI355X") and BASELINE.md section 3's
        # other models -- every rank runs the legs below on its own replicas at the same time, rank 0 reports the SUM of the per-rank rates.
        # Collectives sit OUTSIDE the try blocks and every rank reaches every one of them whatever happens to its own leg: a rank that
        # fails contributes 0 and is counted out in `ranks_measured`; a leg whose set-up fails on any rank is skipped by all of them.
        def all_ranks(setup, measure):
            state, ok = None, 1.0
            try:
                state = setup()
            except Exception as ex:
                ok = 0.0
                sys.stderr.write("rank %d: set-up failed: %r\n" % (rank, ex))
            if env.sum_over_ranks(ok) < world:       # collective 1
                return None
            val, ok = 0.0, 1.0
            env.barrier()                            # collective 2: start together
            try:
                val = measure(state)
            except Exception as ex:
                ok = 0.0
                sys.stderr.write("rank %d: leg failed: %r\n" % (rank, ex))
            return env.sum_over_ranks(val), int(env.sum_over_ranks(ok))   # collectives 3, 4

        holder = {}

        def gather_setup():
            # (--rows-cap: plumbing runs with several ranks on one device, or on the CPU back-end, cannot hold a full replica per rank)
            mcb = fr.Model.builtin(fr.MODEL_C).clone(index_mode=fr.INDEX_PER_BANK, max_rows=args.rows_cap)
            cbk = fr.Context(mcb, device=local_rank)
            cbk.fill_tables(fr.FILL_HASH, SEED_TABLES)
            cbk.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            cbk.synchronize()
            holder["c"] = (mcb, cbk)
            return mcb, cbk

        r_ = all_ranks(gather_setup, lambda st: leg_gather(fr, st[1], st[0], 4096, "uniform", reps=10 if cpu else 200, nbuf=2 if cpu else 32, seed=dist_mod.replica_seed(SEED_IDX, rank))["achieved"])
        leg_done("gather_per_bank_all_ranks")
        if rank == 0 and r_ is not None:
            result["gather_per_bank_all_ranks"] = {"achieved": r_[0], "unit": "GB/s", "peak": HBM_PEAK_GBS * world, "frac": r_[0] / (HBM_PEAK_GBS * world),
                                                   "ranks_measured": r_[1], "bound": "hbm",
                                                   "what": "sum over the ranks of the algorithmic GB/s of fr_worker_gather_only, Model-C batch 4096, one index per bank, "
                                                           "every rank on its own replica at the same time (HIP events on each rank's stream)"}

        def config_rate(ctx_, model_, Bc, prec):
            if cpu:   # rehearsal: the CPU back-end computes in fp32; a token batch and step count
                Bc = 128
                rngc = np.random.default_rng(dist_mod.replica_seed(SEED_IDX + 5, rank))
                ih = [uniform_idx(rngc, model_.index_ranges(), Bc) for _ in range(2)]
                dh = [rngc.uniform(-1, 1, (Bc, model_.dense_len)).astype(np.float32) for _ in range(2)] if model_.dense_len else None
                di = [fr.DeviceBuffer.from_numpy(ctx_, a_) for a_ in ih]
                dd = [fr.DeviceBuffer.from_numpy(ctx_, a_) for a_ in dh] if dh else None
                dv = fr.Driver(ctx_, args.threads, args.depth, Bc)
                el_ = dv.run_resident(Bc, 4, di, dd)
                dv.close()
                return 4 * Bc / el_
            rngc = np.random.default_rng(dist_mod.replica_seed(SEED_IDX + 5, rank))
            ih = [uniform_idx(rngc, model_.index_ranges(), Bc) for _ in range(8)]
            dh = [rngc.uniform(-1, 1, (Bc, model_.dense_len)).astype(np.float32) for _ in range(8)] if model_.dense_len else None
            di = [fr.DeviceBuffer.from_numpy(ctx_, a_) for a_ in ih]
            dd = [fr.DeviceBuffer.from_numpy(ctx_, a_) for a_ in dh] if dh else None
            ctx_.set_fc_precision({"bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[prec])
            if prec == "fp8":
                cal = fr.Worker(ctx_, Bc)
                cal.calibrate_fp8(ih[0], dh[0] if dh else None)
                cal.close()
            dv = fr.Driver(ctx_, args.threads, args.depth, Bc)
            dv.run_resident(Bc, 128, di, dd)
            n_ = steady_run(lambda k: dv.run_resident(Bc, k, di, dd), 1.0, n_first=256, quantum=64)
            el_ = dv.run_resident(Bc, n_, di, dd)
            dv.close()
            for b_ in di + (dd or []):
                b_.free()
            return n_ * Bc / el_

        if not args.no_multi_configs:
            cfg_all = []
            for prec in ("bf16", "fp8"):   # Model-C per-bank replicas (the context of the gather leg)
                r_ = all_ranks(lambda: holder["c"], lambda st, p_=prec: config_rate(st[1], st[0], 4096, p_))
                if rank == 0 and r_ is not None:
                    cfg_all.append({"tag": "C4096_%s_per_bank" % prec, "workload": "Model-C batch=4096, %s FC chain end to end, one index per bank, one replica per rank" % prec, "dtype": prec,
                                    "value": r_[0], "unit": "inferences/s", "ranks_measured": r_[1], "how": "sum of the per-rank rates measured at the same time (>= 1 s each)"})
            if "c" in holder:
                holder["c"][1].close()
                del holder["c"]

            def b_setup():
                mbb = fr.Model.builtin(fr.MODEL_B).clone(max_rows=args.rows_cap)
                cbb = fr.Context(mbb, device=local_rank)
                cbb.fill_tables(fr.FILL_HASH, SEED_TABLES)
                cbb.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
                cbb.set_stream_group(bf16_launch_group(1024))   # as the N = 1 line's configs[2] row
                holder["b"] = (mbb, cbb)
                return mbb, cbb

            r_ = all_ranks(b_setup, lambda st: config_rate(st[1], st[0], 1024, "bf16"))
            if rank == 0 and r_ is not None:
                cfg_all.append({"tag": "B1024_bf16", "workload": "BASELINE configs[2]: Model-B batch=1024, bf16 FC, fused concat + FC chain, one replica per rank", "dtype": "bf16",
                                "value": r_[0], "unit": "inferences/s", "ranks_measured": r_[1], "how": "sum of the per-rank rates measured at the same time (>= 1 s each)"})
            if "b" in holder:
                holder["b"][1].close()
            if rank == 0:
                result["configs_all_ranks"] = cfg_all
        elif "c" in holder:
            holder["c"][1].close()
        leg_done("configs_all_ranks")

    if multi and not args.no_multi_sharded:
        # N > 1: the table-sharded split north_star names (BASELINE configs[3] / [4]) on the DEFAULT line -- Model-C batch 4096, tables
        # sharded by table-ID over the N ranks, one all-gather of the looked-up slices per step (bf16 transport), FC on B / N items per rank;
        # and, when the node holds it, configs[4]: every table 5 x its rows (316 GB: past one GPU's 288 GB), fp8 FC.  Fresh contexts (the
        # replicas above are closed).  Set-up failures are voted on before the first data-path collective; a rank that dies inside the
        # step loop takes the job down through the launcher (self_launch / torchrun stop the other ranks), it cannot hang the line.
        import types
        cases = [("configs[3]", dict(precision="bf16", row_scale=1.0))]
        m5 = fr.Model.builtin(fr.MODEL_C).clone(row_scale=5.0)
        offs5, lens5, _ = m5.shard_plan(world)
        shard5 = max(sum(t.rows * t.dim * 4 for si in m5.segments() if si.kind == fr.SEG_TABLE and offs5[k] <= si.rec_offset < offs5[k] + lens5[k]
                         for t in [m5.tables()[si.src]]) for k in range(world))
        if cpu:   # rehearsal: both configurations' plans (configs[4] with the 5 x inflated row counts under the cap), fp32 chains
            cases = [("configs[3]", dict(precision="f32", row_scale=1.0)), ("configs[4]", dict(precision="f32", row_scale=5.0))]
        elif not args.share_device and m5.min_shards() is not None and m5.min_shards() <= world and shard5 <= 0.85 * 288e9:
            cases.append(("configs[4]", dict(precision="fp8", row_scale=5.0)))
        # These legs have data-path collectives INSIDE their step loop: every rank runs them under a LineGuard (above).
        keys = {"configs[3]": "sharded", "configs[4]": "sharded_inflated_fp8"}
        guard = LineGuard(rank, result)
        for name, kw in cases:
            sa = types.SimpleNamespace(batch=4096, steps=2 if cpu else 50, warmup=1 if cpu else 10, transport="lp", exchange="allgather", backend=args.backend,
                                       rows_cap=args.rows_cap, no_unsharded_check=False, **kw)
            res = guard.run("%s leg (%s)" % (keys[name], name), 300.0, lambda: run_sharded(fr, dist_mod, env, local_rank, sa, auto_steps_s=0.0 if cpu else 1.0))
            res["baseline_config"] = name
            if rank == 0:
                result[keys[name]] = res
            leg_done(keys[name])
        guard.close()
        # ... and the C-ABI's OWN RCCL step (fr_comm_init_rank + fr_worker_submit_sharded) with G = N ranks -- the one piece of the product no
        # one-GPU box can execute -- each rank in a child process, so that it cannot cost the line (a failure is a string in `sharded_cabi_rccl`)
        cabi_env = os.environ.get("FR_BENCH_CABI_RCCL", "1")   # 0: skip; force: also with --share-device (two ranks on one GPU: RCCL refuses -> exercises the failure path)
        if not cpu and (not args.share_device or cabi_env == "force") and env.dist is not None and cabi_env != "0":
            try:
                cres = leg_cabi_rccl(fr, env, local_rank)
            except Exception as ex:   # noqa: BLE001
                cres = {"ok": False, "error": repr(ex)[:300]} if rank == 0 else None
            if rank == 0:
                result["sharded_cabi_rccl"] = cres
            leg_done("sharded_cabi_rccl")

    if rank == 0:
        result["leg_seconds_total"] = round(sum(leg_s.values()), 1)
        sys.stderr.write("bench.py: wall time per leg (s): %s; total %.1f s\n" % (", ".join("%s %.1f" % kv for kv in leg_s.items() if kv[1] >= 0.05), result["leg_seconds_total"]))
        emit(result)
    env.close()


def main_plumbing(args, graft):
    """CPU check of the launcher + rendezvous + timing rule: no GPU, no library.  Every rank 'processes' --steps fake steps."""
    import importlib.util
    fr_dir = graft.PKG_DIR
    spec = importlib.util.spec_from_file_location("fleetrec_dist_only", os.path.join(fr_dir, "dist.py"))
    dist_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dist_mod)
    if args.fail_rank >= 0 and int(os.environ.get("RANK", "0")) == args.fail_rank:
        sys.stderr.write("rank %d: injected failure before the rendezvous\n" % args.fail_rank)
        sys.exit(3)
    env = dist_mod.DistEnv("gloo" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    env.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + env.rank))
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    result = None
    if env.rank == 0:
        result = {"metric": "plumbing-only", "value": None, "unit": "inferences/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                  "data": "none", "config": {"workload": "launcher / rendezvous / max-over-ranks check, no GPU work",
                                             "self_launched": os.environ.get("FR_BENCH_SELF_LAUNCHED") == "1"},
                  # the objects of the real line, empty: the compact-line writer is the same code (tests/test_dist_gloo.py checks the shape)
                  "roofline": {"bound": "none", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None},
                  "cpu_baseline": {"value": None, "unit": "inferences/s", "cores": 0, "kind": "port", "sample": "none (plumbing only)"}}
    if args.fail_sharded_rank >= 0:
        # the collective-leg guard of the N > 1 line, on the CPU: rank --fail-sharded-rank raises inside the "leg", its peers sit in a barrier it
        # never joins.  Expected: the line (with `sharded_error`) still appears on stdout, and the job's exit status is LineGuard.STATUS.
        guard = LineGuard(env.rank, result)

        def leg():
            if env.rank == args.fail_sharded_rank:
                time.sleep(0.5)   # let the peers get into their collective first
                raise RuntimeError("injected failure inside the sharded leg")
            env.barrier()
        guard.run("sharded leg (injected failure)", 60.0, leg)
        guard.close()
    if env.rank == 0:
        emit(result)
    env.close()


if __name__ == "__main__":
    main()
