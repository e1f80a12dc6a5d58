"""Phase timing inside fr_fused_tile_h_kernel (Model-B batch 1024 bf16; diagnostic stamps of wave 0, s_memrealtime at 100 MHz):
one launch of 16 batches = 256 workgroups of 64 items, alone on the chip."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_B)
ctx = fr.Context(m, 0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
ctx.set_fc_precision(fr.FC_BF16)
B, NB = 1024, (int(sys.argv[1]) if len(sys.argv) > 1 else 16)   # NB batches per launch = 16 NB workgroups of 64 items
NWG = NB * B // 64
rng = np.random.default_rng(0)
pool = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)) for _ in range(16)]
sc = [fr.DeviceBuffer(ctx, B * 4) for _ in range(32)]
wk = fr.Worker(ctx, B)
for rep in range(200):
    for i in range(NB):
        wk.push_device(B, pool[i % 16], None, sc[i % 32])
wk.sync()
stamps = fr.DeviceBuffer(ctx, 8 * 4096 * 16 * 8)
stamps.upload(np.zeros(8 * 4096 * 16, np.uint64))
lib = fr.lib()
lib.fr_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.fr_debug_set_stamp_buffer(stamps.ptr)
for i in range(NB):
    wk.push_device(B, pool[i % 16], None, sc[i % 32])
wk.sync()
lib.fr_debug_set_stamp_buffer(None)
sw = stamps.download(np.uint64, 8 * 4096 * 16)[:8 * 16 * NWG].reshape(NWG, 8, 16).astype(np.int64)   # [workgroup][wave][stamp]
s = sw[:, 0, :]
names = ["start", "gather done", "FC1 c0", "FC2 c0", "FC1 c1", "FC2 c1", "FC1 c2", "FC2 c2", "FC1 c3", "FC2 c3", "FC3 + R3"]
t0 = s[:, 0].min()
print("workgroups %d, launch span %.1f us (first start -> last FC3)" % (NWG, (s[:, 10].max() - t0) / 100.0))
prev = 0.0
for i, nme in enumerate(names):
    col = (s[:, i] - s[:, 0]) / 100.0
    print("%-12s median %6.1f us (+%.1f)   max %6.1f" % (nme, np.median(col), np.median(col) - prev, col.max()))
    prev = np.median(col)
print("per wave (median over workgroups, us since the workgroup's wave 0 started): " + "  ".join(names[1:]))
for w in range(8):
    if not (sw[:, w, 1] > 0).any():
        continue
    print("wave %d: " % w + "  ".join("%5.1f" % np.median((sw[:, w, i] - sw[:, 0, 0]) / 100.0) for i in range(1, 11)))
live = sw[:, :, 1] > 0   # the 4-wave kernel stamps waves 0..3 only
clk = ((sw[:, :, 15] - sw[:, :, 14]) / np.maximum(sw[:, :, 9] - sw[:, :, 1], 1) * 0.1)[live]   # shader cycles per 10 ns tick -> GHz
print("in-kernel clock over the FC1 / FC2 chunks (s_memtime / s_memrealtime, median over waves): %.3f GHz; shader cycles there: %.0f (MFMA work per SIMD: 4 x 348 x 32 = 44544)" % (np.median(clk), np.median((sw[:, :, 15] - sw[:, :, 14])[live])))
