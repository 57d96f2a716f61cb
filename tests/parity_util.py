"""Attention-parity bookkeeping: every attention parity assertion goes through check(), which records the OBSERVED
error next to the bound it is held to (gpurun_out/parity_errors.jsonl on the GPU box; summarised into
profiles/rNN_parity_errors.json by tools/parity_summary.py) and then asserts.  The north star's bar is 1e-3 for
fp16 logits; tests name the bound they use and why when it is not that."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl")

# unit roundoff of the output formats: an output of magnitude |o| carries at least u * |o| of rounding error
U16 = {"float16": 2.0 ** -11, "fp16": 2.0 ** -11, "bfloat16": 2.0 ** -8, "bf16": 2.0 ** -8}


def check(err, tol, tag=None):
    """Record (test id, observed max-abs error, bound) and assert err <= tol."""
    err, tol = float(err), float(tol)
    rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "err": err, "tol": tol}
    if tag is not None:
        rec["tag"] = str(tag)
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    assert err <= tol, (tag, err, tol)
    return err


def _ulp(x, mant_bits):
    """Spacing of a binary float format with `mant_bits` stored mantissa bits at magnitude |x| (normal range)."""
    import numpy as np

    ax = np.maximum(np.abs(x), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(ax)) - mant_bits)


def abs_values(v):
    """|v| of an oracle input array (bf16 travels as uint16 bit patterns, fp16 as numpy float16)."""
    import numpy as np

    return (v & np.uint16(0x7FFF)) if v.dtype == np.uint16 else np.abs(v)


def check_out(got, want, dtype, tag=None, ulps=1.0, absw=None):
    """Element-wise bound for a 16-bit attention OUTPUT against the fp64 oracle, pinned to the north star:

        fp16:  |got - want| <= max(1e-3, ulps * ulp_fp16(|want|))     (1e-3 = the north star's bar; one fp16 ulp
                                                                        only exceeds it where |o| >= 2)
        bf16:  |got - want| <= max(4e-3, ulps * ulp_bf16(|want|))     (bf16 ulp: 3.9e-3 in [0.5, 1), 7.8e-3 in [1, 2))

    `absw` (optional, same shape): the oracle's attention output with |V| in place of V, i.e. A = sum_j p_j |v_j|.
    The kernels round the probabilities P to the 16-bit dtype before the PV product (as the reference's kernels
    do: `p.to(v.dtype)`), which perturbs the output by up to u * A (u = 2^-11 / 2^-8) -- more than one ulp of |o|
    where the values cancel (few visible keys: sliding windows, the first causal rows).  With absw the bound is
        max(floor, ulps * ulp(|want|)) + u * A.
    `ulps` > 1 only where the result provably carries more than one 16-bit rounding (partials merged through
    16-bit buffers, fp8 pools): the caller says why.  Records the observed max-abs error and the worst
    error / bound ratio, then asserts ratio <= 1."""
    import numpy as np

    name = {"torch.float16": "fp16", "torch.bfloat16": "bf16"}.get(str(dtype), str(dtype))
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    if name == "fp16":
        bound = np.maximum(1e-3, ulps * _ulp(want, 10))
    elif name == "bf16":
        bound = np.maximum(4e-3, ulps * _ulp(want, 7))
    else:
        raise ValueError(f"check_out: dtype {dtype}")
    if absw is not None:
        bound = bound + U16[name] * np.asarray(absw, dtype=np.float64)
    diff = np.abs(got - want)
    ratio = float((diff / bound).max()) if diff.size else 0.0
    err = float(diff.max()) if diff.size else 0.0
    # reported, not asserted (profiles/rNN_parity_errors.json): the same check against the fp16 bar
    # max(1e-3, ulps * ulp(|want|)) with no absw term -- for bf16 "would it pass without the 4e-3 floor"
    strict = np.maximum(1e-3, ulps * _ulp(want, 10 if name == "fp16" else 7))
    ratio_strict = float((diff / strict).max()) if diff.size else 0.0
    rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "err": err,
           "tol": float(bound.min()) if diff.size else 0.0, "ratio": ratio, "dtype": name, "ulps": ulps,
           "ratio_floor_1e-3_no_absw": ratio_strict, "absw": absw is not None}
    if tag is not None:
        rec["tag"] = str(tag)
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    assert ratio <= 1.0, (tag, name, "max-abs err", err, "worst err/bound", ratio)
    return err


def want_and_absw(fn, args, v_idx, **kwargs):
    """The oracle's output and its |V| twin for check_out's `absw`: fn(*args, **kwargs), and the same call with the
    value tensors (positions `v_idx` of args) replaced by their absolute values."""
    import inspect

    # the oracle's attention functions return the twin from the same softmax in one pass (return_absw); anything else --
    # or a call that also asks for the LSE -- takes the second call
    if "return_absw" in inspect.signature(fn).parameters and not kwargs.get("return_lse"):
        return fn(*args, return_absw=True, **kwargs)
    want = fn(*args, **kwargs)
    a2 = list(args)
    for i in v_idx:
        a2[i] = abs_values(a2[i])
    return want, fn(*a2, **kwargs)
