"""Phase timing inside fr_fused_tile_hk_kernel (the K-outer persistent bf16 fused kernel; Model-B batch 1024 by default): diagnostic
stamps of every wave's lane 0 (s_memrealtime at 100 MHz) of the FIRST tile of every workgroup, and the kernel end.
usage: fused_hk_stamps.py [batches per launch = 16] [model B|A]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
fr = g.load_package()
which = sys.argv[2] if len(sys.argv) > 2 else "B"
m = fr.Model.builtin(fr.MODEL_B if which == "B" else fr.MODEL_A)
ctx = fr.Context(m, 0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
ctx.set_fc_precision(fr.FC_BF16)
B, NB = (1024 if which == "B" else 256), (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
tiles = NB * B // 64
NWG = min(tiles, 256)
rng = np.random.default_rng(0)
pool = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)) for _ in range(16)]
sc = [fr.DeviceBuffer(ctx, B * 4) for _ in range(64)]
wk = fr.Worker(ctx, B)
for rep in range(200):
    for i in range(NB):
        wk.push_device(B, pool[i % 16], None, sc[i % 64])
wk.sync()
stamps = fr.DeviceBuffer(ctx, 8 * 4096 * 16 * 8)
stamps.upload(np.zeros(8 * 4096 * 16, np.uint64))
lib = fr.lib()
lib.fr_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.fr_debug_set_stamp_buffer(stamps.ptr)
for i in range(NB):
    wk.push_device(B, pool[i % 16], None, sc[i % 64])
wk.sync()
lib.fr_debug_set_stamp_buffer(None)
sw = stamps.download(np.uint64, 8 * 4096 * 16)[:8 * 16 * NWG].reshape(NWG, 8, 16).astype(np.int64)   # [workgroup][wave][stamp]
s = sw[:, 0, :]
names = ["start", "prologue (slices 0,1 in LDS)", "FC1 done", "R1 stored", "FC2 done", "FC3 + R3", "scores (tile 0 end)", "kernel end"]
t0 = s[:, 0].min()
print("model %s, %d batches per launch = %d tiles on %d workgroups (%.1f tiles each); launch span %.1f us" % (which, NB, tiles, NWG, tiles / NWG, (s[:, 7].max() - t0) / 100.0))
prev = 0.0
for i, nme in enumerate(names):
    col = (s[:, i] - s[:, 0]) / 100.0
    print("%-30s median %6.1f us (+%.1f)   min %6.1f max %6.1f" % (nme, np.median(col), np.median(col) - prev, col.min(), col.max()))
    prev = np.median(col)
print("per wave (median over workgroups, us since the workgroup's wave 0 started):")
for w in range(8):
    print("wave %d: " % w + "  ".join("%5.1f" % np.median((sw[:, w, i] - sw[:, 0, 0]) / 100.0) for i in range(1, 8)))
clk = (sw[:, :, 15] - sw[:, :, 14]) / np.maximum(sw[:, :, 2] - sw[:, :, 1], 1) * 0.1   # shader cycles per 10 ns tick -> GHz
kg = 55 if which == "B" else 22
print("in-kernel clock over FC1 of tile 0: %.3f GHz; shader cycles there: %.0f (MFMA work per SIMD: 2 waves x %d x 8 x 32 = %d)" % (
    np.median(clk), np.median(sw[:, :, 15] - sw[:, :, 14]), kg, 2 * kg * 8 * 32))
