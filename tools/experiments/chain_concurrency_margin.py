#!/usr/bin/env python3
"""The two figures tests/test_gpu_chain.py::test_chain_workers_run_their_layers_side_by_side asserts on, printed (Model-C 4096 bf16, chain width 4):
(1) an FC1 launch's time per stream with a second worker launching beside it / alone; (2) four chains' rate / one chain's rate."""
import sys, threading, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
B = 4096
rng = np.random.default_rng(66)
idx = (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)
dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
ctx.set_fc_precision(fr.FC_BF16); ctx.set_chain_width(4)
wks = [fr.Worker(ctx, B) for _ in range(4)]
for w in wks: w.infer(idx, dense)
def per_launch_ms(act, reps=60, layer=0):
    for w in act: w.fc_layer_repeat(B, layer, 10)
    for w in act: w.sync()
    stops = [None]*len(act)
    def run_one(i, w):
        w.timer_start(); w.fc_layer_repeat(B, layer, reps); stops[i] = w.timer_stop_ms()
    th = [threading.Thread(target=run_one, args=(i, w)) for i, w in enumerate(act)]
    [t.start() for t in th]; [t.join() for t in th]
    return float(np.mean(stops))/reps
d_i, d_d = fr.DeviceBuffer.from_numpy(ctx, idx), fr.DeviceBuffer.from_numpy(ctx, dense)
d_s = [fr.DeviceBuffer(ctx, B*4) for _ in wks]
def rate(act, n=24):
    for w in act: w.sync()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for w, sc in zip(act, d_s): w.push_device(B, d_i, d_d, sc)
    for w in act: w.sync()
    return len(act)*n*B/(time.perf_counter()-t0)
rate(wks)
for rep in range(4):
    a, p = per_launch_ms(wks[:1]), per_launch_ms(wks[:2])
    r4, r1 = rate(wks), rate(wks[:1])
    f1, f4 = per_launch_ms(wks[:1], layer=1), per_launch_ms(wks, layer=1)
    print("FC2 per launch: alone %.1f us, four workers at once %.1f us -> %.2f kernels resident on average" % (1e3 * f1, 1e3 * f4, 4 * f1 / f4))
    print("FC1 per launch: alone %.1f us, beside a second worker %.1f us (ratio %.3f)   chains: four %.1f M inf/s, one %.1f M (gain %.2f)" % (1e3*a, 1e3*p, p/a, r4/1e6, r1/1e6, r4/r1))
