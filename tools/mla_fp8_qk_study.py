#!/usr/bin/env python3
"""CPU study (no GPU): would an fp8 x fp8 MFMA QK^T -- q quantised to e4m3 inside the MLA decode kernel, the latent
rows used as stored -- stay inside the parity bound of the fp8-row decode path?  

The product path upcasts the e4m3 rows to 16 bits (exact) and runs the bf16 / fp16 MFMA against the 16-bit q: its only
error against the fp64 oracle is the 16-bit rounding of P and of the output.  Here the same attention is evaluated in
fp64 twice -- with q as given, and with q rounded to e4m3 under the best per-(request, head) power-of-two or amax scale --
and the output difference is set against the bound the parity tests use (tests/parity_util.py: 1 ulp of the output
dtype at the output's magnitude, floor 1e-3 relative for fp16 / 4e-3 for bf16).  python tools/mla_fp8_qk_study.py"""
import numpy as np


def e4m3_round(x):
    """Round to the nearest e4m3fn value (3 mantissa bits, exponent bias 7, max 448, subnormals at 2^-9)."""
    x = np.asarray(x, dtype=np.float64)
    s, a = np.sign(x), np.minimum(np.abs(x), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                 # subnormal range shares the exponent -6
    step = 2.0 ** (e - 3)
    return s * np.round(a / step) * step


def study(ctx, heads, seed, dist):
    rng = np.random.default_rng(seed)
    dk, dv = 576, 512
    if dist == "normal":
        kv = rng.standard_normal((ctx, dk))
        q = rng.standard_normal((heads, dk))
    else:  # a few large channels, as rope / outlier dimensions have
        kv = rng.standard_normal((ctx, dk)) * (1 + 7 * (rng.random(dk) < 0.03))
        q = rng.standard_normal((heads, dk)) * (1 + 7 * (rng.random(dk) < 0.03))
    kv = e4m3_round(kv)                       # the pool's rows (exact in both paths)
    sm = 192 ** -0.5

    def attn(qm):
        s = (qm @ kv.T) * sm
        p = np.exp(s - s.max(axis=1, keepdims=True))
        p /= p.sum(axis=1, keepdims=True)
        return p @ kv[:, :dv], s

    ref, s_ref = attn(q)
    amax = np.abs(q).max(axis=1, keepdims=True)
    scale = 448.0 / amax                      # per-head amax scaling: the most accurate choice
    q8 = e4m3_round(q * scale) / scale
    got, s8 = attn(q8)
    out_mag = np.abs(ref).max()
    err = np.abs(got - ref).max()
    return dict(ctx=ctx, dist=dist, logit_rms_err=float(np.sqrt(np.mean((s8 - s_ref) ** 2))), out_err=float(err),
                out_mag=float(out_mag), rel=float(err / out_mag),
                x_fp16_bound=float(err / (1e-3 * out_mag)), x_bf16_bound=float(err / (4e-3 * out_mag)))


if __name__ == "__main__":
    import json
    for dist in ("normal", "outlier_channels"):
        for ctx in (512, 8192):
            print(json.dumps(study(ctx, 16, 1, dist)))
