#!/usr/bin/env python3
"""Model-B 1024 bf16 through the native driver (2 x 2 workers, 128 batches per launch): inferences/s of whatever library FR_LIB names, under
whatever FR_FUSED_HS_ABLATE says (diagnostic build: 1 = the producers load no rows, 2 = every row load reads row 0).  argv[1] = table | bank."""
import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
fr = g.load_package()
B, NB = 1024, 32
mode = sys.argv[1] if len(sys.argv) > 1 else "table"
m = fr.Model.builtin(fr.MODEL_B)
if mode == "bank": m = m.clone(index_mode=fr.INDEX_PER_BANK)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 1); ctx.fill_weights(fr.WEIGHTS_UNIFORM, 2)
rng = np.random.default_rng(66)
rr = m.index_ranges()
d_i = [fr.DeviceBuffer.from_numpy(ctx, (rng.random((B, len(rr))) * rr[None, :]).astype(np.int32)) for _ in range(NB)]
ctx.set_fc_precision(fr.FC_BF16)
ctx.set_stream_group(128)
dv = fr.Driver(ctx, 2, 2, B)
dv.run_resident(B, 1024, d_i, None)
rates = []
for rep in range(3):
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        dv.run_resident(B, 4096, d_i, None); n += 4096
    rates.append(n * B / (time.perf_counter() - t0) / 1e6)
print("%s lib=%s ablate=%s: %s M inf/s" % (mode, os.path.basename(fr.LIB_PATH), os.environ.get("FR_FUSED_HS_ABLATE", "0"), " ".join("%.1f" % r for r in rates)), flush=True)
