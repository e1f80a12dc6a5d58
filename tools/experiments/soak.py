#!/usr/bin/env python3
"""Sustained rate of the headline configuration (Model-A batch 256 fp32, fr_driver_run_resident, 2 x 2 workers): consecutive windows of
~2 s for `argv[1]` seconds (default 30) -- does the 2-s figure on the bench line hold when the chip is warm?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402
import numpy as np  # noqa: E402

fr = graft.load_package()
total_s = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
m = fr.Model.builtin(fr.MODEL_A)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, bench.SEED_TABLES)
ctx.fill_weights(fr.WEIGHTS_UNIFORM, bench.SEED_WEIGHTS)
rng = np.random.default_rng(bench.SEED_IDX)
B = 256
d_idx = [fr.DeviceBuffer.from_numpy(ctx, bench.uniform_idx(rng, m.rows(), B)) for _ in range(64)]
drv = fr.Driver(ctx, 2, 2, B)
drv.run_resident(B, 8192, d_idx)
n = 560000
t_end = time.time() + total_s
w = 0
while time.time() < t_end:
    el = drv.run_resident(B, n, d_idx)
    print("window %2d: %.2f s, %.2f M inferences/s" % (w, el, n * B / el / 1e6), flush=True)
    w += 1
drv.close()
