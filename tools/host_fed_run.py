#!/usr/bin/env python3
"""The two request streams of bench.py's headline side by side, short, for a profiler: Model-A batch 256 fp32, (1) index rows resident in HBM
(fr_driver_run_resident) and (2) the same stream fed from host memory with scores delivered to host memory (fr_driver_run_host_streaming:
the reference loop's H2D / D2H included, cuda_server.c:460-461,494-495).  Prints both rates; phases are separated by a 50 ms pause so that a
timeline can tell them apart.  Usage: host_fed_run.py [threads] [depth] [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

fr = g.load_package()
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
B = 256
m = fr.Model.builtin(fr.MODEL_A)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 0xF1EE7)
ctx.fill_weights(fr.WEIGHTS_UNIFORM, 99)
rng = np.random.default_rng(1234)
host = [(rng.random((B, m.n_tables)) * m.rows()[None, :]).astype(np.int32) for _ in range(64)]
dev = [fr.DeviceBuffer.from_numpy(ctx, a) for a in host]
out = {}
for name, run in (("resident", lambda dv, n: dv.run_resident(B, n, dev)), ("host_fed", lambda dv, n: dv.run_host(B, n, host, streaming=True))):
    dv = fr.Driver(ctx, threads, depth, B)
    run(dv, 8192)
    el = run(dv, 16384)
    n = max(8192, int(16384 / el * secs) // 256 * 256)
    ctx.synchronize()
    time.sleep(0.05)
    t0 = time.time()
    el = run(dv, n)
    out[name] = n * B / el
    print("%-9s %d threads x %d workers: %d batches in %.4f s = %.2f M inferences/s (phase starts at wall %.6f)" % (name, threads, depth, n, el, out[name] / 1e6, t0), flush=True)
    dv.close()
    time.sleep(0.05)
print("host_fed / resident = %.4f" % (out["host_fed"] / out["resident"]))
