#!/usr/bin/env python3
"""Model-C batch-4096 record-producing gather under the experiment knobs of fr_gather.hip (FR_GATHER_XCD, FR_GATHER_ITEMS),
per-table and per-bank index modes, uniform indices.  Run on the GPU box: python tools/experiments/gather_sweep.py"""
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
    for xcd, items, loop in (("1", "4", "1"), ("1", "4", "2"), ("1", "4", "4"), ("1", "2", "2"), ("1", "2", "4"), ("1", "2", "8"), ("1", "8", "2"), ("1", "1", "4"), ("1", "1", "8"),
                             ("0", "8", "1"), ("1", "4", "1")):
        os.environ["FR_GATHER_XCD"], os.environ["FR_GATHER_ITEMS"], os.environ["FR_GATHER_LOOP"] = xcd, items, loop
        r = bench.leg_gather(fr, ctx, m, 4096, "uniform", reps=200, nbuf=32)
        print("%s xcd=%s items=%s loop=%s: %.2f us  %.0f GB/s  frac %.3f" % (name, xcd, items, loop, 1e3 * r["avg_launch_ms"], r["achieved"], r["frac"]), flush=True)
    ctx.close()
