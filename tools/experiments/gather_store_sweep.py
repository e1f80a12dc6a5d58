#!/usr/bin/env python3
"""Model-C per-bank gather at batch 4096 / 8192 against the cache policy of the record stores (FR_GATHER_STORE, read per launch):
plain write-back stores leave up to 32 MiB of dirty L2 lines for the end-of-kernel write-back; write-through (sc1) stores drain while
the kernel still reads.  Interleaved rounds on one box.  Run on the GPU box: python tools/experiments/gather_store_sweep.py"""
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
    for rnd in range(3):
        for B in (4096, 8192):
            for st, label in ((0, "plain"), (16, "sc1"), (2, "nt")):
                os.environ["FR_GATHER_STORE"] = str(st)
                os.environ["FR_GATHER_STREAM"] = "0"   # the one-chunk-per-workgroup form
                r = bench.leg_gather(fr, ctx, m, B, "uniform", reps=200, nbuf=32)
                print("%s round %d batch %5d stores %-8s: %.2f us  %.0f GB/s algorithmic  frac %.3f" % (name, rnd, B, label, 1e3 * r["avg_launch_ms"], r["achieved"], r["frac"]), flush=True)
    os.environ.pop("FR_GATHER_STORE", None)
    ctx.close()
