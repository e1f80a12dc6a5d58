#!/usr/bin/env python3
"""Race screen of the GEMM-layer chain (Model-C batch 4096, chain width 4: fc_pp_gemm_kernel on FC1 / FC2, fc_lp_gemm_out_kernel, the gather) under
its real neighbours: four workers on four host threads push device-resident batches for `seconds`; every delivered score vector is compared BIT FOR
BIT with the one a lone worker computed for the same rows before the soak.  A DMA that lands late, a stage overwritten early or a barrier that
does not cover a read shows as a flipped score (such races come and go with what else runs on the CU).  workers = 1: chain width 1, the lone worker's
full-chip tiles (fc_pp_gemm_n128_kernel).  `bank` (round 6): a FR_INDEX_PER_BANK context -- the reference scores are computed with the gather reading
the fp32 rows (fr_ctx_set_lp_bank_image(0)), the soak runs over the operand-type bank image: a race in gather_tr_stream_lp_body's LDS tile AND any
difference between the two row sources show as a flipped score.  Usage: soak_chain.py [seconds] [bf16|fp8] [workers: 4] [batch: 4096] [table|bank]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

fr = g.load_package()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n_workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
per_bank = len(sys.argv) > 5 and sys.argv[5] == "bank"
m = fr.Model.builtin(fr.MODEL_C)
if per_bank:
    m = m.clone(index_mode=fr.INDEX_PER_BANK)
ctx = fr.Context(m, device=0)
ctx.fill_tables(fr.FILL_HASH, 0xF1EE7)
ctx.fill_weights(fr.WEIGHTS_UNIFORM, 99)
ctx.set_fc_precision({"bf16": fr.FC_BF16, "fp8": fr.FC_FP8}[prec])
ctx.set_chain_width(min(4, n_workers))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
rng0 = np.random.default_rng(1)
NP = 6
pool, d_pool = [], []
ref_wk = fr.Worker(ctx, B)
for j in range(NP):
    rr = m.index_ranges()
    idx = (rng0.random((B, len(rr))) * rr[None, :]).astype(np.int32)
    dense = rng0.uniform(-1, 1, (B, m.dense_len)).astype(np.float32)
    if prec == "fp8" and j == 0:
        ref_wk.calibrate_fp8(idx, dense)
    if per_bank:
        ctx.set_lp_bank_image(0)   # the references: fp32 rows, converted by the gather
    pool.append(ref_wk.infer(idx, dense).copy())
    d_pool.append((fr.DeviceBuffer.from_numpy(ctx, idx), fr.DeviceBuffer.from_numpy(ctx, dense)))
if per_bank:
    ctx.set_lp_bank_image(1)       # the soak: rows already in the operand type
ref_wk.fc_layer_only(B, 0)
k0 = ref_wk.last_kernel()
ref_wk.sync()
ref_wk.close()
stop = time.time() + secs
errors, counts = [], [0] * n_workers


def run(t):
    rng = np.random.default_rng(100 + t)
    wk = fr.Worker(ctx, B)
    outs = [fr.DeviceBuffer(ctx, B * 4) for _ in range(8)]
    try:
        while time.time() < stop and not errors:
            ks = [int(rng.integers(0, NP)) for _ in range(8)]
            for o, k in zip(outs, ks):
                wk.submit_device(B, d_pool[k][0], d_pool[k][1], o)
                wk.sync()
                got = o.download(np.float32, B)
                if not np.array_equal(got, pool[k]):
                    bad = np.flatnonzero(got != pool[k])
                    errors.append("thread %d batch %d: %d scores differ, first at %d: %r vs %r" % (t, k, len(bad), bad[0], got[bad[0]], pool[k][bad[0]]))
                    break
                counts[t] += 1
    except Exception as ex:   # noqa: BLE001
        errors.append("thread %d: %r" % (t, ex))
    wk.close()


th = [threading.Thread(target=run, args=(t,)) for t in range(n_workers)]
for t_ in th:
    t_.start()
for t_ in th:
    t_.join()
print("soak_chain %s%s: FC1 = %s, %d batches of %d in %.0f s on %d worker(s), %d mismatches" % (prec, " per bank (operand-type image %.2f GB)" % (ctx.lp_bank_image_bytes() / 1e9) if per_bank else "", k0, sum(counts), B, secs, n_workers, len(errors)))
for e in errors[:5]:
    print("  ", e)
sys.exit(1 if errors else 0)
