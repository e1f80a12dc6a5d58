#!/usr/bin/env python3
"""One configuration of the operand-type bank image A/B under rocprofv3 --kernel-trace --stats: argv = precision (bf16|fp8), image (0|1), mode
(lone: one worker's stream of pushes | chains: four chains through the native driver).  Per-bank Model-C, batch 4096."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
prec, on, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
m = fr.Model.builtin(fr.MODEL_C).clone(index_mode=fr.INDEX_PER_BANK)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
B = 4096
rng = np.random.default_rng(66)
_, brows = m.bank_map()
NB = 16
idx = [(rng.random((B, len(brows))) * brows[None, :]).astype(np.int32) for _ in range(NB)]
dense = [rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) for _ in range(NB)]
d_i = [fr.DeviceBuffer.from_numpy(ctx, a) for a in idx]
d_d = [fr.DeviceBuffer.from_numpy(ctx, a) for a in dense]
ctx.set_fc_precision({"bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[prec])
ctx.set_chain_width(4)
if prec == "fp8":
    cal = fr.Worker(ctx, B); cal.calibrate_fp8(idx[0], dense[0]); cal.close()
ctx.set_lp_bank_image(on)
if mode == "lone":
    wk = fr.Worker(ctx, B)
    d_s = fr.DeviceBuffer(ctx, B * 4)
    for k in range(400): wk.push_device(B, d_i[k % NB], d_d[k % NB], d_s)
    wk.sync()
else:
    dv = fr.Driver(ctx, 2, 2, B)
    dv.run_resident(B, 2048, d_i, d_d)
    dv.close()
print("done", prec, on, mode)
