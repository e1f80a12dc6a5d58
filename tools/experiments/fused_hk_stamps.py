"""Barrier-by-barrier timing inside fr_fused_tile_hs_kernel (the K-outer, persistent, wave-specialised bf16 fused kernel): every wave's
lane 0 stamps s_memrealtime (100 MHz) when it ARRIVES at each workgroup barrier of the workgroup's first two tiles and when the barrier
RELEASES it.  Who arrives last says who the step waited for: the consumers (MFMA waves 0-7) or the producers (gather waves 8-11).
The stamps exist in the DIAGNOSTIC build only: FR_LIB=<package>/libfleetrec_diag.so (make -C csrc diag); that build's kernel is ~17 % slower
than the product's (spilled registers), so read the pattern, not the absolute times.
usage: FR_LIB=.../libfleetrec_diag.so fused_hk_stamps.py [batches per launch = 16] [model B|A]"""
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
NSL = 8 if which == "B" else 6
tiles = NB * B // 64
NWG = min(tiles, 256)
rng = np.random.default_rng(0)
pool = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)) for _ in range(16)]
sc = [fr.DeviceBuffer(ctx, B * 4) for _ in range(64)]
if NB > 64:
    ctx.set_stream_group(min(NB, 256))   # one launch carries the whole list (the persistent kernel reads it from device memory)
wk = fr.Worker(ctx, B)
for rep in range(200):
    for i in range(NB):
        wk.push_device(B, pool[i % 16], None, sc[i % 64])
wk.sync()
NST = 12 * 256 * 128
stamps = fr.DeviceBuffer(ctx, NST * 8)
stamps.upload(np.zeros(NST, np.uint64))
lib = fr.lib()
lib.fr_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.fr_debug_set_stamp_buffer(stamps.ptr)
for i in range(NB):
    wk.push_device(B, pool[i % 16], None, sc[i % 64])
wk.sync()
lib.fr_debug_set_stamp_buffer(None)
sw = stamps.download(np.uint64, NST)[:12 * 128 * NWG].reshape(NWG, 12, 128).astype(np.int64)   # [workgroup][wave][slot]
NBAR = NSL + 5
t0 = sw[:, 8:, 0].min(axis=1)                      # the workgroup's first wave start
rel = lambda x: (x - t0[:, None]) / 100.0          # us since the workgroup started
end = rel(sw[:, 8:, 126])
print("model %s, %d batches per launch = %d tiles on %d workgroups (%.1f tiles each); launch span %.1f us, workgroup end median %.1f us" % (
    which, NB, tiles, NWG, tiles / NWG, (sw[:, 8:, 126].max() - sw[:, 8:, 0].min()) / 100.0, np.median(end.max(axis=1))))
cons = bool(sw[:, :8, 126].any())   # the consumers stamp only in a -DFR_STAMP_CONSUMERS build (their code is then not the product's)
if cons:
    print("set-up done (descriptors in LDS, rings / prologue): consumers %.1f us" % np.median(rel(sw[:, :8, 1]).max(axis=1)))
names = ["slice %d" % s for s in range(NSL)] + ["R1 stored", "FC2 done", "R2 stored", "R3 stored", "partials"]
for tile in range(2):
    if tiles / NWG <= tile:
        break
    print("tile %d: barrier            %sproducers arrive   released   producers waited   (median over workgroups of the LAST wave of each role, us)" % (tile, "consumers arrive   " if cons else ""))
    prev = None
    for b, nme in enumerate(names):
        arr = rel(sw[:, :, 4 + 2 * NBAR * tile + 2 * b]); out = rel(sw[:, :, 5 + 2 * NBAR * tile + 2 * b])
        pa, ro = np.median(arr[:, 8:].max(axis=1)), np.median(out[:, 8:].max(axis=1))
        wait = np.median(out[:, 8:].max(axis=1) - arr[:, 8:].max(axis=1))
        ctext = ""
        if cons:
            ca = np.median(arr[:, :8].max(axis=1))
            ctext = "%8.1f %s        " % (ca, "*" if ca >= pa else " ")
        print("        %-18s %s%8.1f        %8.1f   %8.1f       %s" % (nme, ctext, pa, ro, wait, "" if prev is None else "(+%.1f)" % (ro - prev)))
        prev = ro
if cons:
    cyc = (sw[:, :8, 3] - sw[:, :8, 2]).astype(np.float64)
    fc1 = (sw[:, :8, 4 + 2 * NSL] - sw[:, :8, 5]).astype(np.float64)   # release of barrier 0 .. arrival at "R1 stored" of tile 0 (10 ns ticks)
    kg = 55 if which == "B" else 22
    print("in-kernel clock over FC1 of tile 0: %.3f GHz; shader cycles there: %.0f (MFMA work per SIMD: 2 waves x %d x 8 x 32 = %d)" % (
        np.median(cyc / np.maximum(fc1, 1) * 0.1), np.median(cyc), kg, 2 * kg * 8 * 32))

