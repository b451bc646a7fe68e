#!/usr/bin/env python3
"""Generates tests/golden/*.npz + golden.json.

The reference (enp1s0/tsqr-gpu) cannot be built or run in this image and ships no vectors, so these fixtures
hold: the seeded input, |R| (sign-normalised) from the oracle restatement for the five restated modes (the two half-typed ones on
the fp16 rounding of the same input), |R| from LAPACK in
fp64 (numpy), and residual / orthogonality bands.  They pin the oracle (and the input generator) against drift;
they are not reference output.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_oracle as ro  # noqa: E402

cases = []
for (m, n, seed) in [(128, 16, 0), (9211, 51, 0), (4096, 64, 0)]:
    a = ro.uniform_matrix(m, n, seed=seed)
    out = {"a_head": np.asarray(a[:8, :4]).copy()}
    resmax, orthmax = {}, {}
    for name, md in (("fp32_notc", ro.FP32_NOTC), ("fp32_tc_cor", ro.FP32_TC_COR), ("fp32_tc_nocor", ro.FP32_TC_NOCOR),
                     ("fp16_notc", ro.FP16_NOTC), ("fp16_tc_nocor", ro.FP16_TC_NOCOR)):
        st, q, r = ro.qr(a, md, False)
        assert st == 0
        out["absr_" + name] = np.abs(np.triu(r)).astype(np.float32)
        ain = a.astype(np.float16).astype(np.float32) if name.startswith("fp16") else a     # (io type half: the input those modes see)
        resmax[name] = float(3 * ro.residual(ain, q, r))
        orthmax[name] = float(3 * ro.orthogonality_fro(q))
    out["absr_lapack64"] = np.abs(np.linalg.qr(a.astype(np.float64), mode="r")).astype(np.float32)
    fn = "uniform_%dx%d_seed%d.npz" % (m, n, seed)
    np.savez_compressed(os.path.join(HERE, fn), **out)
    cases.append({"file": fn, "m": m, "n": n, "seed": seed, "a_sha256": hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(), "residual_max": resmax, "orth_fro_max": orthmax})
json.dump({"generator": "tests/golden/make_golden.py", "cases": cases}, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
print("wrote", [c["file"] for c in cases])
