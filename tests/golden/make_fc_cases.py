#!/usr/bin/env python3
"""Writes tests/golden/fc_cases_{A,B,C}.npz: committed FC fixtures -- random-data cases of the 4-GEMM chain
(cuda_server.c:211-217 layout, :468-491 chain: R1 = W1 X, R2 = W2 R1, R3 = W3 R2, out = Wout R3; alpha = 1, beta = 0, no bias, no
activation) with expected scores accumulated in float64.

Inputs of a case (all reproducible from small committed data + stated rules, nothing of the reference's text):
  * index rows + dense features: the first 16 rows of the committed tests/golden/records_*.bin (fixed per-table indices);
  * tables: the procedural hash fill (FR_FILL_HASH, seed 0xF1EE7; oracle content_rows == the device's fill_table_kernel), gathered
    into records by the CPU oracle's bank-addressed gather;
  * weights: W ~ U(-1, 1) / sqrt(K_layer), fp32, column-major H x K (element (h, k) at [h + k H]), from the stated hash rule below
    (== the device's fill_weights_kernel(FR_WEIGHTS_UNIFORM, seed 99), restated here in numpy);
  * expected: float64 chain of the fp32 inputs, no intermediate rounding.
The .npz holds idx, dense, expected, the seeds and SHA-256 digests of the records and of every weight matrix (so that a test that
regenerates them knows it regenerated the same bytes).  cuBLASLt's own fp32 summation order is unknowable (closed library): these
fixtures pin OUR chain to committed numbers, within BASELINE.json's 1e-3; they are not outputs of the reference.
Run from the repo root:  python tests/golden/make_fc_cases.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import oracle as O  # noqa: E402
import make_records  # noqa: E402

SEED_TABLES, SEED_WEIGHTS, N_CASES = 0xF1EE7, 99, 16


def fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def uniform_weights(seed, layer, K, H):
    """W[h + k H] of layer `layer` (K inputs, H outputs): ((int32)(hash >> 8) * 2^-23 - 1) / sqrt(K), all in fp32."""
    n = K * H
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h0 = fmix32(np.array([(seed ^ (((layer + 1) * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF], dtype=np.uint32))[0]
        h = fmix32(np.uint32(h0) ^ (i & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        h = fmix32(h ^ (i >> np.uint64(32)).astype(np.uint32))
    v = (h >> np.uint32(8)).astype(np.int32).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    return (v * (np.float32(1.0) / np.sqrt(np.float32(K)))).astype(np.float32)


def chain_f64(rec_f32, ws, fc):
    x = rec_f32.astype(np.float64).T                       # K x B (item-major records are column-major K x B)
    for l in range(4):
        K, H = fc[l], fc[l + 1]
        W = ws[l].astype(np.float64).reshape(K, H).T       # element (h, k) at [h + k H]
        x = W @ x
    return x[0]


def build(which):
    om = O.OracleModel(which)
    idx, dense, _ = make_records.read(os.path.join(HERE, make_records.FILES[which]))
    idx, dense = np.ascontiguousarray(idx[:N_CASES]), np.ascontiguousarray(dense[:N_CASES])
    rec = om.gather(idx, dense=dense if om.dense_len else None, content_mode=O.FILL_HASH, seed=SEED_TABLES).view(np.float32)
    fc = [int(v) for v in om.fc]
    ws = [uniform_weights(SEED_WEIGHTS, l, fc[l], fc[l + 1]) for l in range(4)]
    return dict(idx=idx.astype(np.int32), dense=dense.astype(np.float32), expected=chain_f64(rec, ws, fc), fc=np.array(fc, dtype=np.int32),
                seed_tables=np.uint32(SEED_TABLES), seed_weights=np.uint32(SEED_WEIGHTS),
                records_sha256=hashlib.sha256(rec.tobytes()).hexdigest(), weights_sha256=np.array([hashlib.sha256(w.tobytes()).hexdigest() for w in ws]))


if __name__ == "__main__":
    for which in "ABC":
        d = build(which)
        path = os.path.join(HERE, "fc_cases_%s.npz" % which)
        np.savez(path, **d)
        print(os.path.basename(path), "expected[:3] =", d["expected"][:3], os.path.getsize(path), "bytes")
