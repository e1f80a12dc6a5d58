"""Model-C batch-4096 chain throughput (2 x 2 workers) as a function of what ran before it in the process: precision order on one context,
an idle second context with its own driver.  usage: python tools/experiments/chain_order_check.py <case>"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
fr = g.load_package()
case = sys.argv[1] if len(sys.argv) > 1 else "fp8"
B = 4096
m = fr.Model.builtin(fr.MODEL_C)
other = None
if "withA" in case:
    ma = fr.Model.builtin(fr.MODEL_A)
    ca = fr.Context(ma, device=0); ca.fill_tables(fr.FILL_HASH, 1); ca.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    other = fr.Driver(ca, 2, 2, 256)
for tag, which, bsz in (("preA", fr.MODEL_A, 256), ("preB", fr.MODEL_B, 1024)):   # a fused-kernel model served and closed BEFORE Model-C
    if tag in case:
        mp = fr.Model.builtin(which)
        cp = fr.Context(mp, device=0); cp.fill_tables(fr.FILL_HASH, 1); cp.fill_weights(fr.WEIGHTS_UNIFORM, 2)
        rp = np.random.default_rng(1)
        ip = [fr.DeviceBuffer.from_numpy(cp, (rp.random((bsz, mp.n_tables)) * mp.rows()[None, :]).astype(np.int32)) for _ in range(8)]
        for prec_ in ((fr.FC_FP32, fr.FC_BF16, fr.FC_FP8) if "all" in case else (fr.FC_FP32,)):
            cp.set_fc_precision(prec_)
            if prec_ == fr.FC_FP8:
                cal = fr.Worker(cp, bsz); cal.calibrate_fp8((rp.random((bsz, mp.n_tables)) * mp.rows()[None, :]).astype(np.int32), None); cal.close()
            dvp = fr.Driver(cp, 2, 2, bsz)
            el = dvp.run_resident(bsz, 20000, ip, None)
            dvp.close()
            print("%s: %s %.1f M inf/s" % (case, tag, 20000 * bsz / el / 1e6), flush=True)
        for b_ in ip:
            b_.free()
        cp.close()
ctx = fr.Context(m, device=0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
rng = np.random.default_rng(5)
ih = [(rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32) for _ in range(8)]
dh = [rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) for _ in range(8)]
di = [fr.DeviceBuffer.from_numpy(ctx, a) for a in ih]
dd = [fr.DeviceBuffer.from_numpy(ctx, a) for a in dh]
P = {"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}
order = [p for p in case.split("_") if p in P]
for rep in range(2):
    for prec in order:
        ctx.set_fc_precision(P[prec])
        if prec == "fp8":
            cal = fr.Worker(ctx, B); cal.calibrate_fp8(ih[0], dh[0]); cal.close()
        dv = fr.Driver(ctx, 2, 2, B)
        dv.run_resident(B, 256, di, dd)
        n = 4096 if prec != "f32" else 512
        el = dv.run_resident(B, n, di, dd)
        dv.close()
        print("%s rep %d %s: %.2f M inf/s" % (case, rep, prec, n * B / el / 1e6), flush=True)
