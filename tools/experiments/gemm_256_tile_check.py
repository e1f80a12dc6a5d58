import sys, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
rng = np.random.default_rng(5)
B = 8192
idx = (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)
dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
for prec, P, tol in (("bf16", fr.FC_BF16, 1e-2), ("fp8", fr.FC_FP8, 4e-2)):
    ctx.set_fc_precision(P)
    wk = fr.Worker(ctx, B)
    if prec == "fp8": wk.calibrate_fp8(idx[:4096], dense[:4096])
    big = wk.infer(idx, dense)
    k1 = wk.last_kernel()
    lo = wk.infer(idx[:4096], dense[:4096]); hi = wk.infer(idx[4096:], dense[4096:])
    ref = np.concatenate([lo, hi])
    err = np.abs(big - ref).max() / np.abs(ref).max()
    print(prec, "batch 8192 vs two batches of 4096: rel err %.2e" % err, "deterministic", np.array_equal(wk.infer(idx, dense), big))
    assert err <= tol
    wk.close()
print("ok")
