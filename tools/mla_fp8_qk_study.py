#!/usr/bin/env python3
"""CPU study (no GPU): can the fp8-row MLA decode run its two contractions on the fp8 MFMA -- latent rows fed to the
matrix core as stored, no upcast -- and stay inside the parity bound of the fp8-row decode path?  (VERDICT r02 item 8,
r04 item 4.)

The product path upcasts the e4m3 rows to 16 bits (exact) and runs the bf16 / fp16 MFMA against the 16-bit q: its only
error against the fp64 oracle is the 16-bit rounding of P and of the output.  Here the same attention is evaluated in
fp64 with the operands an fp8 x fp8 MFMA would see:

  q    ONE e4m3 rounding under a per-head amax scale (round 3's question: 35-116 x the bound -- no), or a SUM of
       2 / 3 e4m3 terms (hi + lo [+ lo2], each term the rounding of what the terms before it left over), each with the
       block scale `v_mfma_scale_f32_16x16x128_f8f6f4` gives it for free: one power of two per 32 contraction elements
       (E8M0).  A term costs one fp8 MFMA = half a bf16 MFMA's cycles.
  P    likewise 1 / 2 / 3 e4m3 terms of exp(s - m), block scale per 32 keys.

and the output difference is set against the bound the parity tests use (tests/parity_util.py check_out: max(floor,
1 ulp of the output dtype at |o|) + u * A, A = sum_j p_j |v_j|, u = 2^-8 bf16 / 2^-11 fp16 -- the term that pays for the
16-bit rounding of P in the product path), evaluated element by element like the tests do.  q is first rounded to the
16-bit dtype it arrives in.   python tools/mla_fp8_qk_study.py"""
import json

import numpy as np


def e4m3_round(x):
    """Round to the nearest e4m3fn value (3 mantissa bits, exponent bias 7, max 448, subnormals at 2^-9)."""
    x = np.asarray(x, dtype=np.float64)
    s, a = np.sign(x), np.minimum(np.abs(x), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                 # subnormal range shares the exponent -6
    step = 2.0 ** (e - 3)
    return s * np.round(a / step) * step


def round16(x, dtype):
    mant = 7 if dtype == "bf16" else 10
    x = np.asarray(x, dtype=np.float64)
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -24 if dtype == "fp16" else 2.0 ** -126)))
    if dtype == "fp16":
        e = np.maximum(e, -14.0)
    step = 2.0 ** (e - mant)
    return np.round(x / step) * step


def fp8_terms(x, terms, block=32):
    """x [rows, K] as a sum of `terms` e4m3 tensors, each with one power-of-two scale per `block` elements of K (E8M0: the
    scale that puts the block's amax just under 448)."""
    rows, k = x.shape
    pad = (-k) % block
    xp = np.pad(x, ((0, 0), (0, pad))).reshape(rows, -1, block)
    total = np.zeros_like(xp)
    for _ in range(terms):
        r = xp - total
        amax = np.abs(r).max(axis=2, keepdims=True)
        sc = 2.0 ** np.floor(np.log2(448.0 / np.maximum(amax, 2.0 ** -60)))
        total = total + e4m3_round(r * sc) / sc
    return total.reshape(rows, -1)[:, :k]


def study(ctx, heads, seed, dist, dtype, q_terms, p_terms):
    rng = np.random.default_rng(seed)
    dk, dv = 576, 512
    if dist == "normal":
        kv = rng.standard_normal((ctx, dk))
        q = rng.standard_normal((heads, dk))
    else:  # a few large channels, as rope / outlier dimensions have
        kv = rng.standard_normal((ctx, dk)) * (1 + 7 * (rng.random(dk) < 0.03))
        q = rng.standard_normal((heads, dk)) * (1 + 7 * (rng.random(dk) < 0.03))
    kv = e4m3_round(kv)                       # the pool's rows (exact in both paths)
    q = round16(q, dtype)                     # q as it arrives
    sm = 192 ** -0.5
    s_ref = (q @ kv.T) * sm
    p_ref = np.exp(s_ref - s_ref.max(axis=1, keepdims=True))
    ref = (p_ref @ kv[:, :dv]) / p_ref.sum(axis=1, keepdims=True)
    absw = (p_ref @ np.abs(kv[:, :dv])) / p_ref.sum(axis=1, keepdims=True)

    q8 = fp8_terms(q, q_terms) if q_terms else q
    s = (q8 @ kv.T) * sm
    p = np.exp(s - s.max(axis=1, keepdims=True))
    l = p.sum(axis=1, keepdims=True)          # the row sum stays in fp32 in the kernel
    if p_terms:
        p8 = fp8_terms(p, p_terms)            # blocks of 32 KEYS per head row
    else:
        p8 = round16(p, dtype)                # the product path: P rounded to the 16-bit dtype
    got = round16((p8 @ kv[:, :dv]) / l, dtype)
    u = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    mant = 7 if dtype == "bf16" else 10
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - mant)
    bound = np.maximum(4e-3 if dtype == "bf16" else 1e-3, ulp) + u * absw
    ratio = np.abs(got - ref) / bound
    return dict(ctx=ctx, dist=dist, dtype=dtype, q_terms=q_terms or "16-bit", p_terms=p_terms or "16-bit",
                logit_rms_err=float(np.sqrt(np.mean((s - s_ref) ** 2))), worst_err_over_bound=float(ratio.max()),
                mean_err_over_bound=float(ratio.mean()))


if __name__ == "__main__":
    rows = []
    for dtype in ("bf16", "fp16"):
        for dist in ("normal", "outlier_channels"):
            for ctx in (512, 8192):
                for qt, pt in ((0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (0, 2), (0, 3), (2, 2), (3, 2), (3, 3)):
                    r = study(ctx, 16, 1, dist, dtype, qt, pt)
                    rows.append(r)
                    print(json.dumps(r))
    print()
    print("worst err / bound over the four inputs, per dtype and (q terms, P terms):")
    for dtype in ("bf16", "fp16"):
        for qt, pt in ((0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (0, 2), (0, 3), (2, 2), (3, 2), (3, 3)):
            w = max(r["worst_err_over_bound"] for r in rows if r["dtype"] == dtype and r["q_terms"] == (qt or "16-bit") and r["p_terms"] == (pt or "16-bit"))
            print(f"  {dtype}  q {qt or '16-bit':>6}  P {pt or '16-bit':>6}  ->  {w:8.3f}")
