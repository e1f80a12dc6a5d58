"""Scores of Model-C batch 4096 (bf16, fp8) on the 256 x 256 GEMM tile of the running library build -> an .npz; with two files: compare bit
for bit.  The experiments build chooses the kernel by FR_LP_GEMM_PP (0 = fc_lp_gemm_kernel, 2 / 3 = fc_pp_gemm_kernel<., D>), once per
process, hence one process per variant:  FR_LIB=...exp.so FR_LP_GEMM_PP=3 python gemm_pp_check.py out3.npz ; python gemm_pp_check.py out0.npz out3.npz"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    ok = True
    for k in a.files:
        if k.startswith("kernel"):
            print(k, str(a[k]), "|", str(b[k]))
            continue
        same = np.array_equal(a[k], b[k])
        print(k, "bit-identical" if same else "DIFFERENT: max rel %.3e" % (np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()))
        ok &= same
    sys.exit(0 if ok else 1)
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
ctx.set_chain_width(int(os.environ.get("FR_CHECK_WIDTH", "4")))   # 1: the full-chip 128 x 256 tiles (FR_LP_GEMM_PP128: fc_pp_gemm_n128_kernel)
rng = np.random.default_rng(5)
out = {}
for B in (4096, 8192):
    idx = (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)
    dense = rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    for prec, P in (("bf16", fr.FC_BF16), ("fp8", fr.FC_FP8)):
        ctx.set_fc_precision(P)
        wk = fr.Worker(ctx, B)
        if prec == "fp8": wk.calibrate_fp8(idx[:4096], dense[:4096])
        s1 = wk.infer(idx, dense)
        for _ in range(20):
            assert np.array_equal(wk.infer(idx, dense), s1), "not deterministic"
        out["%s_%d" % (prec, B)] = s1
        wk.fc_layer_only(B, 0); wk.sync()
        out["kernel_%s_%d" % (prec, B)] = np.array(wk.last_kernel())
        wk.close()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], {k: str(v) for k, v in out.items() if k.startswith("kernel")})
