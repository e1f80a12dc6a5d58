#!/usr/bin/env python3
"""A/B of the operand-type bank image (fr_ctx_set_lp_bank_image) on the per-bank Model-C chain at batch 4096:
(1) the chain's gather launch alone on one stream (HIP events), image on / off; (2) four chains through the native driver, image on / off."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
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
for prec, enum in (("bf16", fr.FC_BF16), ("fp8", fr.FC_FP8)):
    ctx.set_fc_precision(enum)
    ctx.set_chain_width(4)
    if prec == "fp8":
        cal = fr.Worker(ctx, B); cal.calibrate_fp8(idx[0], dense[0]); cal.close()
    for on in (0, 1, 0, 1):
        ctx.set_lp_bank_image(on)
        dv = fr.Driver(ctx, 2, 2, B)
        dv.run_resident(B, 64, d_i, d_d)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.5:
            dv.run_resident(B, 256, d_i, d_d); n += 256
        dt = time.perf_counter() - t0
        dv.close()
        # the gather launch alone: one worker, pushes of the gather stage only are not exposed -> time a lone worker's stream of pushes instead
        wk = fr.Worker(ctx, B)
        d_s = fr.DeviceBuffer(ctx, B * 4)
        for k in range(8): wk.push_device(B, d_i[k % NB], d_d[k % NB], d_s)
        wk.sync()
        wk.timer_start()
        for k in range(64): wk.push_device(B, d_i[k % NB], d_d[k % NB], d_s)
        ms = wk.timer_stop_ms()
        wk.close(); d_s.free()
        print("%s image %d: four chains %.2f M inf/s; one lone worker %.1f us per batch; image bytes %.2f GB" % (prec, on, n * B / dt / 1e6, 1e3 * ms / 64, ctx.lp_bank_image_bytes() / 1e9), flush=True)
