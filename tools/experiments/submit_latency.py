#!/usr/bin/env python3
"""fr_worker_submit + fr_worker_sync latency from the host against the batch size (Model-A fp32), p50 / p90 of 400 calls.
FR_SUBMIT_ZEROCOPY=0/1 (read when the library loads) selects copy commands vs kernels reading the pinned buffers over PCIe."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402
fr = graft.load_package()
m = fr.Model.builtin(fr.MODEL_A)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, bench.SEED_TABLES)
ctx.fill_weights(fr.WEIGHTS_UNIFORM, bench.SEED_WEIGHTS)
rng = np.random.default_rng(1)
for B in (1, 8, 32, 64, 128, 256, 1024):
    wk = fr.Worker(ctx, B)
    idx = bench.uniform_idx(rng, m.rows(), B)
    for _ in range(50):
        wk.infer(idx)
    ts = []
    for _ in range(400):
        t0 = time.perf_counter()
        wk.infer(idx)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("zero-copy %s batch %4d: p50 %.1f us  p90 %.1f us" % (os.environ.get("FR_SUBMIT_ZEROCOPY", "0"), B, 1e6 * ts[200], 1e6 * ts[360]), flush=True)
    wk.close()
