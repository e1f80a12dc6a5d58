#!/usr/bin/env python3
"""The software-pipelined gather (FR_GATHER_STREAM chunks per workgroup, write-through stores) against the one-chunk-per-workgroup form
over batch sizes, Model-C, both index laws.  Run on the GPU box: python tools/experiments/gather_stream_batches.py"""
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
        row = []
        for nstep, st in ((0, 0), (1, 16), (2, 16), (4, 16), (8, 16), (2, 0)):
            os.environ["FR_GATHER_STREAM"] = str(nstep)
            os.environ["FR_GATHER_STORE"] = str(st)
            r = bench.leg_gather(fr, ctx, m, B, "uniform", reps=100, nbuf=16)
            row.append("%s%d: %.2f us (%.3f)" % ("wt" if st else "wb", nstep, 1e3 * r["avg_launch_ms"], r["frac"]))
        print("%s batch %5d | %s" % (name, B, " | ".join(row)), flush=True)
    for k in ("FR_GATHER_STREAM", "FR_GATHER_STORE"):
        os.environ.pop(k, None)
    ctx.close()
