#!/usr/bin/env python3
"""Which single-layer launch makes the cleanest witness of 'four workers = four hardware queues'?  Model-C bf16, chain width 4: per layer
and batch size, the time per launch alone and with all four workers launching at once -> average kernels resident = 4 x alone / together."""
import sys, threading
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
ctx.set_fc_precision(fr.FC_BF16); ctx.set_chain_width(4)
rng = np.random.default_rng(66)
wks = [fr.Worker(ctx, 4096) for _ in range(4)]
def per_launch_ms(act, B, layer, reps=60):
    for w in act: w.fc_layer_repeat(B, layer, 10)
    for w in act: w.sync()
    stops = [None]*len(act)
    def run_one(i, w):
        w.timer_start(); w.fc_layer_repeat(B, layer, reps); stops[i] = w.timer_stop_ms()
    th = [threading.Thread(target=run_one, args=(i, w)) for i, w in enumerate(act)]
    [t.start() for t in th]; [t.join() for t in th]
    return float(np.mean(stops))/reps
for B in (4096, 2048, 1024):
    idx = (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    for w in wks: w.infer(idx, dense)
    for layer in (0, 1, 2):
        wks[0].fc_layer_only(B, layer); k = wks[0].last_kernel(); wks[0].sync()
        for rep in range(3):
            a = per_launch_ms(wks[:1], B, layer); p2 = per_launch_ms(wks[:2], B, layer); p4 = per_launch_ms(wks, B, layer)
            print("B %4d layer %d (%s): alone %.1f us, two %.1f us (resident %.2f of 2), four %.1f us (resident %.2f of 4)" % (B, layer, k[:40], 1e3*a, 1e3*p2, 2*a/p2, 1e3*p4, 4*a/p4), flush=True)
