#!/usr/bin/env python3
"""Model-C batch-4096 record-producing gather: XCD partition of the record words (uniform n_words / 8 vs cut on source rows, with
different cost weights -- FR_GATHER_COST = "write,l2,cache,hbm", read when the context is created).
Run on the GPU box: python tools/experiments/gather_groups_sweep.py [per_bank|per_table|both]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

fr = graft.load_package()
mc = fr.Model.builtin(fr.MODEL_C)
which = sys.argv[1] if len(sys.argv) > 1 else "per_bank"
modes = [(fr.INDEX_PER_BANK, "per_bank"), (fr.INDEX_PER_TABLE, "per_table")]
for mode, name in modes:
    if which not in (name, "both"):
        continue
    m = mc.clone(index_mode=mode)
    for cost in ("1,0,0,0", "1,0.25,0.5,1", "1,0,0.5,1", "1,0.5,1,2", "0,0,0,1", "1,1,1,1"):
        os.environ["FR_GATHER_COST"] = cost
        ctx = fr.Context(m, device=0)
        ctx.fill_tables(fr.FILL_HASH, bench.SEED_TABLES)
        st = ctx.gather_groups()
        for uni in ("1", "0", "1", "0") if cost == "1,0,0,0" else ("0", "0"):
            os.environ["FR_GATHER_UNIFORM_GROUPS"] = uni
            r = bench.leg_gather(fr, ctx, m, 4096, "uniform", reps=200, nbuf=32)
            print("%s cost=%s uniform=%s groups=%s: %.2f us  %.0f GB/s  frac %.3f" % (name, cost, uni, [st[g + 1] - st[g] for g in range(8)], 1e3 * r["avg_launch_ms"], r["achieved"], r["frac"]), flush=True)
        ctx.close()
