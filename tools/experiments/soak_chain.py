#!/usr/bin/env python3
"""Sustained rate of the Model-C batch-4096 chain (fr_driver_run_resident, 2 x 2 workers on their own hardware queues, part-chip GEMM tiles):
consecutive windows of ~2 s for `argv[2]` seconds (default 60) in precision `argv[1]` (bf16 | fp8 | f32), the driver re-created every fifth
window (worker streams destroyed and created again), scores of one batch compared with the first window's after every window."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402
import numpy as np  # noqa: E402

fr = graft.load_package()
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
total_s = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
m = fr.Model.builtin(fr.MODEL_C)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, bench.SEED_TABLES)
ctx.fill_weights(fr.WEIGHTS_UNIFORM, bench.SEED_WEIGHTS)
rng = np.random.default_rng(bench.SEED_IDX)
B = 4096
ih = [bench.uniform_idx(rng, m.rows(), B) for _ in range(8)]
dh = [rng.uniform(-1, 1, (B, m.dense_len)).astype(np.float32) for _ in range(8)]
di = [fr.DeviceBuffer.from_numpy(ctx, a) for a in ih]
dd = [fr.DeviceBuffer.from_numpy(ctx, a) for a in dh]
ctx.set_fc_precision({"f32": fr.FC_FP32, "bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[prec])
if prec == "fp8":
    cal = fr.Worker(ctx, B)
    cal.calibrate_fp8(ih[0], dh[0])
    cal.close()
drv = fr.Driver(ctx, 2, 2, B)
drv.run_resident(B, 512, di, dd)
n = {"f32": 2048, "bf16": 16384, "fp8": 24576}[prec]
t_end = time.time() + total_s
w, ref = 0, None
while time.time() < t_end:
    el = drv.run_resident(B, n, di, dd)
    chk = fr.Worker(ctx, B)          # a fifth worker for the check: the rule's divisor stays 4
    got = chk.infer(ih[0], dh[0])
    chk.close()
    if ref is None:
        ref = got
    same = bool(np.array_equal(got, ref))
    print("window %2d: %.2f s, %.2f M inferences/s, check batch identical to the first window's: %s" % (w, el, n * B / el / 1e6, same), flush=True)
    if not same:
        sys.exit("scores changed")
    w += 1
    if w % 5 == 0:
        drv.close()
        drv = fr.Driver(ctx, 2, 2, B)
drv.close()
print("ok: %d windows" % w)
