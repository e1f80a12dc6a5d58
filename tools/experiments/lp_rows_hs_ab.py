#!/usr/bin/env python3
"""A/B of operand-type (bf16) rows under the persistent bf16 fused kernel (fr_fused_tile_hs_kernel): Model-B batch 1024, 128 batches per launch,
per-table and per-bank contexts; image on (SRC = 1, four row sets in flight) / off (fp32 rows, two)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
B, NB = 1024, 32
for mode in ("table", "bank"):
    m = fr.Model.builtin(fr.MODEL_B)
    if mode == "bank": m = m.clone(index_mode=fr.INDEX_PER_BANK)
    ctx = fr.Context(m, device=0)
    ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
    rng = np.random.default_rng(66)
    rr = m.index_ranges()
    d_i = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, len(rr))) * rr[None, :]).astype(np.int32)) for _ in range(NB)]
    ctx.set_fc_precision(fr.FC_BF16)
    ctx.set_stream_group(128)
    for on in (0, 1, 0, 1):
        ctx.set_lp_bank_image(on)
        dv = fr.Driver(ctx, 2, 2, B)
        dv.run_resident(B, 1024, d_i, None)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.5:
            dv.run_resident(B, 4096, d_i, None); n += 4096
        dt = time.perf_counter() - t0
        dv.close()
        wk = fr.Worker(ctx, B)
        d_s = fr.DeviceBuffer(ctx, B * 4)
        for k in range(256): wk.push_device(B, d_i[k % NB], None, d_s)
        wk.sync()
        wk.timer_start()
        for k in range(1280): wk.push_device(B, d_i[k % NB], None, d_s)
        ms = wk.timer_stop_ms()
        kern = wk.last_kernel()
        wk.close(); d_s.free()
        print("B1024 bf16 per-%s rows-image %d: driver %.1f M inf/s; one stream %.1f us per launch of 128 batches (%s)" % (mode, on, n * B / dt / 1e6, 1e3 * ms / 10, kern), flush=True)
    ctx.close()
