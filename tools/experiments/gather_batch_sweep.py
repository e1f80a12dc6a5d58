#!/usr/bin/env python3
"""Model-C record-producing gather (fr_worker_gather_only) against the batch size: how much of a batch-4096 launch is ramp-up and tail.
Run on the GPU box: python tools/experiments/gather_batch_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

fr = graft.load_package()
mc = fr.Model.builtin(fr.MODEL_C)
for mode, name in ((fr.INDEX_PER_BANK, "per_bank"), (fr.INDEX_PER_TABLE, "per_table")):
    m = mc.clone(index_mode=mode)
    ctx = fr.Context(m, device=0)
    ctx.fill_tables(fr.FILL_HASH, bench.SEED_TABLES)
    for B in (1024, 2048, 4096, 8192, 16384, 32768):
        r = bench.leg_gather(fr, ctx, m, B, "uniform", reps=100, nbuf=16)
        print("%s batch %5d: %.2f us  %.0f GB/s algorithmic  frac %.3f  (%.2f ns per item)" % (name, B, 1e3 * r["avg_launch_ms"], r["achieved"], r["frac"], 1e6 * r["avg_launch_ms"] / B), flush=True)
    ctx.close()
