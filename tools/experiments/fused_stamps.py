"""Phase timing inside fr_fused_tile_kernel (diagnostic stamps, s_memrealtime at 100 MHz): one launch of NBATCH (argv[1], default 32) batches of 256 = 8 * NBATCH workgroups, alone on the chip."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_A)
ctx = fr.Context(m, 0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
B = 256
NBATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NWG = 8 * NBATCH
rng = np.random.default_rng(0)
pool = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)) for _ in range(8)]
sc = [fr.DeviceBuffer(ctx, B * 4) for _ in range(16)]
ctx.set_stream_group(NBATCH)
wk = fr.Worker(ctx, B)
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 4   # launches before the stamped one (clocks ramp over many launches)
for rep in range(WARM):
    for i in range(NBATCH):
        wk.push_device(B, pool[i % 8], None, sc[i % 16])
    wk.sync()
NB = 16 * 1024
stamps = fr.DeviceBuffer(ctx, NB * 16 * 8)
stamps.upload(np.zeros(NB * 16, np.uint64))
lib = fr.lib()
lib.fr_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.fr_debug_set_stamp_buffer(stamps.ptr)
for i in range(NBATCH):
    wk.push_device(B, pool[i % 8], None, sc[i % 16])
wk.sync()
lib.fr_debug_set_stamp_buffer(None)
raw = stamps.download(np.uint64, NB * 16)
s = raw[:16 * NWG].reshape(NWG, 16)
t0 = s[:, 0].min()
names = ["start", "gather done", "FC1 c0", "R1 c0 ready", "FC1 c1 (incl FC2 c0)", "R1 c1", "FC1 c2", "R1 c2", "FC1 c3", "R1 c3", "FC2 done+R2", "FC3+R3", "end"]
print("workgroups: %d, launch span %.1f us" % (len(s), (s[:, 12].max() - t0) / 100.0))
prev = None
for i, nme in enumerate(names):
    col = (s[:, i] - s[:, 0]) / 100.0
    d = "" if prev is None else "  (+%.1f)" % (np.median(col) - prev)
    print("%-24s median %7.1f us  max %7.1f%s" % (nme, np.median(col), col.max(), d))
    prev = np.median(col)
cyc = (s[:, 15] - s[:, 14]).astype(np.float64)
rt = (s[:, 11] - s[:, 1]).astype(np.float64) / 100.0
print("shader clock between 'gather done' and 'FC3+R3': median %.0f MHz (min %.0f, max %.0f)" % (np.median(cyc / rt), (cyc / rt).min(), (cyc / rt).max()))
s2 = raw[16 * NWG: 16 * NWG + NWG * 8 * 4].reshape(NWG, 8, 4).astype(np.int64)
start = s[:, 0:1].astype(np.int64)
for k, nme in enumerate(["FC1(c1) done / barrier arrival", "barrier released", "FC2(c1) done", "FC1(c2) done"]):
    print("per-wave %-32s" % nme, np.round(np.median((s2[:, :, k] - start) / 100.0, axis=0), 1))
