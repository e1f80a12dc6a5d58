#!/usr/bin/env python3
"""Soak of the host-fed streaming path (fr_worker_push_host: copy stream + pinned scores) against fr_worker_submit: random batch sizes, random
flush / poll / sync points, workers created and destroyed along the way, several threads at once; every delivered score is compared bit for bit
with the per-batch submit of the same rows.  Usage: soak_host_fed.py [seconds] [threads]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

fr = g.load_package()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = fr.Model.builtin(fr.MODEL_A)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 0xF1EE7)
ctx.fill_weights(fr.WEIGHTS_UNIFORM, 99)
rows = m.rows()
ref_wk = fr.Worker(ctx, 256)
rng0 = np.random.default_rng(1)
pool = []
for b in (256, 256, 200, 64, 1, 255, 128, 256):
    idx = (rng0.random((b, m.n_tables)) * rows[None, :]).astype(np.int32)
    pool.append((idx, ref_wk.infer(idx)))
# the fused kernel sums in another order than the stage pipeline of submit: the reference for streaming is streaming itself, once, checked against submit to 1e-5
wk0 = fr.Worker(ctx, 256)
stream_ref = []
for idx, sub in pool:
    out = np.empty(len(idx), np.float32)
    wk0.push_host(idx, None, out)
    wk0.sync()
    assert np.abs(out - sub).max() <= 1e-5 * np.abs(sub).max()
    stream_ref.append(out.copy())
wk0.close()
stop = time.time() + secs
errors, counts = [], [0] * n_threads


def run(t):
    rng = np.random.default_rng(100 + t)
    try:
        while time.time() < stop:
            wk = fr.Worker(ctx, 256)
            outs = []
            for _ in range(int(rng.integers(1, 400))):
                k = int(rng.integers(0, len(pool)))
                out = np.full(len(pool[k][0]), np.nan, np.float32)
                wk.push_host(pool[k][0], None, out)
                outs.append((k, out))
                r = rng.random()
                if r < 0.02:
                    wk.flush()
                elif r < 0.04:
                    wk.host_poll()
                elif r < 0.05:
                    wk.sync()
                    for kk, o in outs:
                        if not np.array_equal(o, stream_ref[kk]):
                            raise AssertionError("thread %d: scores differ after a mid-stream sync (pool %d)" % (t, kk))
                    counts[t] += len(outs)
                    outs = []
            wk.sync()
            for kk, o in outs:
                if not np.array_equal(o, stream_ref[kk]):
                    raise AssertionError("thread %d: scores differ (pool %d)" % (t, kk))
            counts[t] += len(outs)
            wk.close()
    except Exception as ex:   # noqa: BLE001
        errors.append(repr(ex))


th = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
[x.start() for x in th]
[x.join() for x in th]
print("soak: %d threads, %.0f s, %d batches checked bit for bit, errors: %s" % (n_threads, secs, sum(counts), errors or "none"))
sys.exit(1 if errors else 0)
