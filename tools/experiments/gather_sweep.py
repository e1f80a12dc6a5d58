"""Where does the Model-C batch-4096 gather lose its bandwidth?  Sweep the table row cap (smaller tables ->
L2 / Infinity-Cache resident) and report algorithmic GB/s.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
fr = g.load_package()

def bytes_per_inf(m):
    rows = sum(s.len * 4 for s in m.segments() if s.kind != fr.SEG_DENSE)
    return rows + 4 * m.n_tables + 4 * m.dense_len + 4 * m.record_len

B = 4096
for cap in (10_000, 100_000, 1_000_000, 10_000_000, 0):
    m = fr.Model.builtin(fr.MODEL_C)
    if cap:
        m = m.clone(max_rows=cap)
    ctx = fr.Context(m, device=0)
    t0 = time.perf_counter(); ctx.fill_tables(fr.FILL_HASH, 1); tf = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    rows = m.rows()
    nbuf = 8
    idxs = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, m.n_tables)) * rows[None, :]).astype(np.int32)) for _ in range(nbuf)]
    dns = [fr.DeviceBuffer.from_numpy(ctx, rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)) for _ in range(nbuf)]
    wk = fr.Worker(ctx, B)
    rec = wk.records_dptr()
    for i in range(10):
        wk.gather_only(B, idxs[i % nbuf], dns[i % nbuf], rec)
    wk.sync()
    reps = 100
    wk.timer_start()
    for i in range(reps):
        wk.gather_only(B, idxs[i % nbuf], dns[i % nbuf], rec)
    ms = wk.timer_stop_ms() / reps
    wk.sync()
    gb = bytes_per_inf(m) * B
    print("cap=%9d tables=%.2f GB fill=%.1f ms (%.2f TB/s)  gather %.2f us -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        cap, m.table_bytes() / 1e9, tf * 1e3, m.table_bytes() / tf / 1e12, ms * 1e3, gb / ms / 1e6, gb / ms / 1e6 / 80))
    wk.close(); ctx.close()
