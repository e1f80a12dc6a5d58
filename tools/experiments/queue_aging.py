#!/usr/bin/env python3
"""Does the history of a process's streams cost a chain model its side-by-side workers?  Model-C bf16, chain width 4, four workers: FC2 residency
(4 x alone / four at once) and the four-chain rate in a FRESH process, then after `aging` rounds of creating / destroying workers and contexts
(as a test suite or a long-lived server does), four new workers each time."""
import sys, threading, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
ctx.set_fc_precision(fr.FC_BF16); ctx.set_chain_width(4)
B = 4096
rng = np.random.default_rng(66)
idx = (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)
dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
d_i, d_d = fr.DeviceBuffer.from_numpy(ctx, idx), fr.DeviceBuffer.from_numpy(ctx, dense)
def per_launch_ms(act, layer, reps=60):
    for w in act: w.fc_layer_repeat(B, layer, 10)
    for w in act: w.sync()
    stops = [None]*len(act)
    def run_one(i, w):
        w.timer_start(); w.fc_layer_repeat(B, layer, reps); stops[i] = w.timer_stop_ms()
    th = [threading.Thread(target=run_one, args=(i, w)) for i, w in enumerate(act)]
    [t.start() for t in th]; [t.join() for t in th]
    return float(np.mean(stops))/reps
def measure(tag):
    wks = [fr.Worker(ctx, B) for _ in range(4)]
    d_s = [fr.DeviceBuffer(ctx, B*4) for _ in wks]
    for w in wks: w.infer(idx, dense)
    res = max(4 * per_launch_ms(wks[:1], 1) / per_launch_ms(wks, 1) for _ in range(3))
    pair = [2 * per_launch_ms(wks[i:i+1], 1) / per_launch_ms([wks[i], wks[j]], 1) for i in range(4) for j in range(i+1, 4)]
    def rate(n=48):
        for w in wks: w.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            for w, sc in zip(wks, d_s): w.push_device(B, d_i, d_d, sc)
        for w in wks: w.sync()
        return 4*n*B/(time.perf_counter()-t0)/1e6
    rate(8)
    r = max(rate() for _ in range(3))
    print("%-28s FC2 resident %.2f of 4; pairs (01 02 03 12 13 23) %s; four chains %.1f M inf/s" % (tag, res, " ".join("%.2f" % p for p in pair), r), flush=True)
    import ctypes, os
    if os.path.basename(fr.LIB_PATH) == "libfleetrec_exp.so":   # does a generic spin burst see the pair that takes turns?
        L = ctypes.CDLL(fr.LIB_PATH)
        L.fr_exp_burst_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_float)] * 2
        for wgs, lds, ticks in ((32, 131072, 3000), (32, 0, 3000), (1, 0, 3000)):
            out = []
            for i in range(4):
                for j in range(i + 1, 4):
                    a, b = ctypes.c_float(), ctypes.c_float()
                    L.fr_exp_burst_probe(wks[i]._h, wks[j]._h, 12, wgs, lds, ticks, ctypes.byref(a), ctypes.byref(b))
                    out.append("%.2f" % (2 * a.value / b.value))
            print("    spin burst (%d wgs, %d KiB LDS, %d us): pair residency %s" % (wgs, lds // 1024, ticks // 100, " ".join(out)), flush=True)
    for w in wks: w.close()
    for x in d_s: x.free()
measure("fresh process")
measure("fresh process, again")
ma = fr.Model.builtin(fr.MODEL_A)
for aging in range(4):
    ca = fr.Context(ma, device=0); ca.fill_tables(fr.FILL_HASH, 1); ca.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    tmp = [fr.Worker(ca, 256) for _ in range(3 + aging)]
    ia = np.zeros((256, ma.n_tables), np.int32)
    for w in tmp: w.infer(ia)
    extra = [fr.Worker(ctx, B) for _ in range(1 + aging % 3)]
    for w in tmp[::2]: w.close()
    measure("after aging round %d" % aging)
    for w in tmp[1::2]: w.close()
    for w in extra: w.close()
    ca.close()
measure("after all aging")
