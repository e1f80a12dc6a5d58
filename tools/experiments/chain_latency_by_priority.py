"""Model-C batch-4096 submit + sync latency of the first worker (highest stream priority) and the second (lowest), each alone on the chip,
then both together; bf16.  usage: python tools/experiments/chain_latency_by_priority.py"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
ctx.set_fc_precision(fr.FC_BF16)
rng = np.random.default_rng(3)
for B in (256, 4096):
    idx = (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    wk = [fr.Worker(ctx, B) for _ in range(2)]
    def lat(w, n=200):
        out = []
        for _ in range(n):
            t0 = time.perf_counter(); w.infer(idx, dense); out.append(time.perf_counter() - t0)
        return 1e6 * float(np.median(out)), 1e6 * float(np.percentile(out, 99))
    for i, w in enumerate(wk):
        lat(w, 20)
        print("batch %4d worker %d (%s priority) alone: p50 %.1f us  p99 %.1f us" % (B, i, "highest" if i == 0 else "lowest", *lat(w)), flush=True)
    res = [None, None]
    th = [threading.Thread(target=lambda i=i: res.__setitem__(i, lat(wk[i], 400))) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(2):
        print("batch %4d worker %d together: p50 %.1f us  p99 %.1f us" % (B, i, *res[i]), flush=True)
    for w in wk: w.close()
