"""In-kernel stamps of one steady-state pipelined launch: when does each stage's workgroups start and finish?
(diagnostic build aid: fr_debug_set_stamp_buffer; s_memrealtime ticks at 100 MHz)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
fr = g.load_package()
m = fr.Model.builtin(fr.MODEL_A)
ctx = fr.Context(m, 0); ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
B = 256
rng = np.random.default_rng(0)
pool = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32)) for _ in range(8)]
sc = [fr.DeviceBuffer(ctx, B * 4) for _ in range(8)]
wk = fr.Worker(ctx, B)
for i in range(50):
    wk.push_device(B, pool[i % 8], None, sc[i % 8])
wk.sync()
NB = 4096
stamps = fr.DeviceBuffer(ctx, NB * 32)
stamps.upload(np.zeros(NB * 4, np.uint64))
lib = fr.lib()
lib.fr_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
for i in range(8):  # fill the pipeline, last launch is steady state
    wk.push_device(B, pool[i % 8], None, sc[i % 8])
ctx.synchronize()
lib.fr_debug_set_stamp_buffer(stamps.ptr)
wk.push_device(B, pool[0], None, sc[0])
ctx.synchronize()
lib.fr_debug_set_stamp_buffer(None)
wk.sync()
s = stamps.download(np.uint64, NB * 4).reshape(NB, 4)
s = s[s[:, 1] > 0]
t0 = s[:, 0].min()
print("workgroups stamped: %d, launch span %.2f us" % (len(s), (s[:, 1].max() - t0) / 100.0))
for st in range(5):
    r = s[s[:, 2] == st]
    if len(r) == 0: continue
    st_us = (r[:, 0] - t0) / 100.0; en_us = (r[:, 1] - t0) / 100.0; du = en_us - st_us
    print("stage %d: %4d WGs  start p0 %.2f p50 %.2f p100 %.2f | end p50 %.2f p100 %.2f | duration p10 %.2f p50 %.2f p90 %.2f p100 %.2f us" % (
        st, len(r), st_us.min(), np.median(st_us), st_us.max(), np.median(en_us), en_us.max(), np.percentile(du, 10), np.median(du), np.percentile(du, 90), du.max()))
xcc = s[:, 3] & 0xF
print("blockIdx%8 -> XCC id agreement:", {int(k): np.bincount((xcc[np.arange(len(s)) % 8 == k]).astype(int), minlength=8).tolist() for k in range(8)})
