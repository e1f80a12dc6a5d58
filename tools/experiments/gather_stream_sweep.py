#!/usr/bin/env python3
"""Model-C gather at batch 4096: the software-pipelined form (FR_GATHER_STREAM = chunks per workgroup, 0 = one-chunk-per-workgroup form)
against items per chunk (FR_GATHER_ITEMS) and the store policy (FR_GATHER_STORE 16 = write-through).  All knobs are read per launch;
interleaved rounds on one box.  Run on the GPU box: python tools/experiments/gather_stream_sweep.py [batches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

fr = graft.load_package()
mc = fr.Model.builtin(fr.MODEL_C)
batches = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096]
for mode, name in ((fr.INDEX_PER_BANK, "per_bank"), (fr.INDEX_PER_TABLE, "per_table")):
    m = mc.clone(index_mode=mode)
    ctx = fr.Context(m, device=0)
    ctx.fill_tables(fr.FILL_HASH, bench.SEED_TABLES)
    for rnd in range(2):
        for B in batches:
            for items in (4, 2):
                for nstep in (0, 1, 2, 4, 8):
                    for st in (0, 16):
                        os.environ["FR_GATHER_ITEMS"] = str(items)
                        os.environ["FR_GATHER_STREAM"] = str(nstep)
                        os.environ["FR_GATHER_STORE"] = str(st)
                        r = bench.leg_gather(fr, ctx, m, B, "uniform", reps=200, nbuf=32)
                        print("%s round %d batch %5d items %d chunks/wg %d stores %-5s: %.2f us  %.0f GB/s algorithmic  frac %.3f" % (
                            name, rnd, B, items, nstep, "sc1" if st else "plain", 1e3 * r["avg_launch_ms"], r["achieved"], r["frac"]), flush=True)
    for k in ("FR_GATHER_ITEMS", "FR_GATHER_STREAM", "FR_GATHER_STORE"):
        os.environ.pop(k, None)
    ctx.close()
