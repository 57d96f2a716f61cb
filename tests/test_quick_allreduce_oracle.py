"""C3 on the CPU: the quick all-reduce's oracle (oracle/radix_oracle.py quick_allreduce) against the properties the
reference's own test asserts (test/manual/test_quick_allreduce.py) and against the definition of its codecs, and the host
class's size gate / environment switches (quick_all_reduce.py:176-244).  The kernel itself is HIP-only: the bit-for-bit
comparison runs in tests/test_gpu_quick_allreduce.py."""
import os

import numpy as np
import pytest
import torch

from oracle import radix_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rcp_f16_gfx950.npy")
TABLE = np.load(GOLDEN) if os.path.exists(GOLDEN) else None   # v_rcp_f16 as an MI355X returns it (tools/dump_rcp_f16.py)


def _bits(x, bf16):
    return O.f32_to_bf16(x.astype(np.float32)) if bf16 else x.astype(np.float16).view(np.uint16)


def _vals(b, bf16):
    return O.bf16_to_f32(b).astype(np.float64) if bf16 else b.view(np.float16).astype(np.float64)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("level", [O.QR_FP, O.QR_INT8, O.QR_INT6, O.QR_INT4])
@pytest.mark.parametrize("bf16,cast", [(False, False), (True, True), (True, False)])
def test_reference_test_properties(world, level, bf16, cast):
    """test_quick_allreduce.py:139-160 (integers in [1, 23): atol 1.25 W, rtol 0.5 W; FP exact), :213-240 (zeros -> zeros,
    ones -> W), :283-300 (a constant v on every rank -> v * W exactly at FP)."""
    rng = np.random.default_rng(world * 10 + level)
    n = 16384 + 40
    ints = [rng.integers(1, 23, n) for _ in range(world)]
    out = _vals(O.quick_allreduce([_bits(i, bf16) for i in ints], bf16, level, cast, rcp_f16_table=TABLE), bf16)
    exact = sum(ints).astype(np.float64)
    err = np.abs(out - exact)
    assert (err <= 1.25 * world + 0.5 * world * np.abs(exact)).all(), err.max()
    if level == O.QR_FP:
        assert err.max() == 0
    ones = O.quick_allreduce([_bits(np.ones(n), bf16)] * world, bf16, level, cast, rcp_f16_table=TABLE)
    assert (_vals(ones, bf16) == world).all()
    zeros = O.quick_allreduce([np.zeros(n, np.uint16)] * world, bf16, level, cast, rcp_f16_table=TABLE)
    assert (zeros & 0x7FFF).max() == 0
    for v in (1, 2, 3, 10):
        c = O.quick_allreduce([_bits(np.full(256, float(v)), bf16)] * world, bf16, O.QR_FP, cast, rcp_f16_table=TABLE)
        assert (_vals(c, bf16) == v * world).all()


@pytest.mark.parametrize("bits", [8, 6, 4])
@pytest.mark.parametrize("bf16", [False, True])
def test_codec_definition(bits, bf16):
    """One codec pass on random activations: codes fill [0, 2^bits) with the block's extreme value on code 0 (the scale is
    -extreme / R: quick_all_reduce.cuh:107-121), the two scale blocks of a 64-element group are its even and its odd
    elements, and the round trip is within half a step (plus the type's roundings) of the input, one more where the clamp at R - 1 bites."""
    rng = np.random.default_rng(bits)
    x = rng.standard_normal(64 * 50)
    xb = _bits(x, bf16)
    num = O._QrBF16() if bf16 else O._QrF16(TABLE)
    dec, code = O.qr_codec_roundtrip(num, xb, bits)
    r = 1 << (bits - 1)
    assert code.min() == 0 and code.max() <= 2 * r - 1
    xv, dv = _vals(xb, bf16).reshape(-1, 32, 2), _vals(dec, bf16).reshape(-1, 32, 2)
    mx, mn = xv.max(axis=1), xv.min(axis=1)
    ext = np.where(np.abs(mx) > np.abs(mn), mx, mn)     # [group, parity]: the maximum if it is larger in magnitude, else the minimum
    step = np.abs(ext) / r
    assert (code.reshape(-1, 32, 2)[xv == ext[:, None, :]] == 0).all()   # the extreme (with its sign) encodes as -R
    # half a step from the rint, plus the 16-bit roundings of x * e, of 1 / d and of the decode product: R * 2^-8 each in bf16
    # (an INT8 code in bf16 is only good to a step), R * 2^-11 in fp16; a full step more where the clamp at R - 1 bites
    slack = 0.5 + 3 * r * 2.0 ** (-8 if bf16 else -11) + 0.02
    err = np.abs(dv - xv) / step[:, None, :]
    clamped = xv * np.sign(-ext)[:, None, :] > (r - 1.5) * step[:, None, :]
    assert err[~clamped].max() <= slack and err.max() <= slack + 1.0, (err[~clamped].max(), err.max(), slack)
    # changing an odd element leaves every even element's round trip alone (its scale block is the other one)
    x2 = x.copy()
    x2[1::2] *= 3.0
    dec2, _ = O.qr_codec_roundtrip(num, _bits(x2, bf16), bits)
    assert np.array_equal(dec2[0::2], dec[0::2])


def test_message_tail_is_zero_padded_per_group():
    """A message that ends inside a 64-element group: the missing elements count as zeros (the reference's bounded buffer
    loads), so the result equals that of the explicitly padded message."""
    rng = np.random.default_rng(5)
    for n in (8, 40, 64 + 24):
        parts = [_bits(rng.standard_normal(n), False) for _ in range(4)]
        padded = [np.concatenate([p, np.zeros(128 - n, np.uint16)]) for p in parts]
        for level in (O.QR_INT8, O.QR_INT4):
            a = O.quick_allreduce(parts, False, level, rcp_f16_table=TABLE)
            b = O.quick_allreduce(padded, False, level, rcp_f16_table=TABLE)
            assert np.array_equal(a, b[:n])


def test_rcp_table_against_correct_rounding():
    """The committed hardware table (when present) is within 1 ulp of the correctly rounded reciprocal -- what the ISA
    promises for v_rcp_f16 -- on every finite non-zero normal input whose reciprocal is a normal number."""
    if TABLE is None:
        pytest.skip("tests/golden/rcp_f16_gfx950.npy not generated yet (tools/dump_rcp_f16.py on the GPU box)")
    cr = O.rcp_f16_correctly_rounded()
    x = np.arange(65536, dtype=np.uint16)
    e = (x >> 10) & 0x1F
    normal = (e > 0) & (e < 31)
    re = (cr >> 10) & 0x1F
    sel = normal & (re > 0) & (re < 31)
    d = np.abs(TABLE[sel].astype(np.int32) - cr[sel].astype(np.int32))
    assert d.max() <= 1


def _gate(world, regime, use_fp16=1, max_mb=0):
    from sglang_amd.parallel import QuickAllReduce, QuickReduceRegime

    q = QuickAllReduce.__new__(QuickAllReduce)
    q.disabled, q.world_size = False, world
    q.qr_quant_level, q.use_fp16_kernels = QuickReduceRegime[regime], use_fp16
    q.qr_max_size = max_mb * (1 << 20) if max_mb > 0 else 1 << 31
    return q


def test_size_gate_follows_the_reference_table():
    """should_quick_allreduce (quick_all_reduce.py:222-244): the level's minimum for (dtype as it travels, world), the
    maximum, 16-byte multiples, 16-bit dtypes."""
    MB = 1 << 20
    q = _gate(8, "INT4")
    assert q.size_ok(torch.float16, 2 * MB) and not q.size_ok(torch.float16, 2 * MB - 16)
    assert q.size_ok(torch.bfloat16, 2 * MB)                       # bf16 travels as fp16 by default: the fp16 row applies
    assert not _gate(8, "INT4", use_fp16=0).size_ok(torch.bfloat16, 1024 * MB)   # native bf16 at world 8: 2 GiB minimum
    assert _gate(2, "FP", use_fp16=0).size_ok(torch.bfloat16, 2 * MB)
    assert not q.size_ok(torch.float32, 4 * MB) and not q.size_ok(torch.float16, 2 * MB + 8)
    assert not _gate(8, "INT4", max_mb=1).size_ok(torch.float16, 2 * MB)
    assert _gate(4, "INT8").size_ok(torch.float16, 16 * MB) and not _gate(4, "INT8").size_ok(torch.float16, 8 * MB)


def test_environment_switches(monkeypatch):
    """NONE (the default) and an unknown level leave the communicator disabled; a group of one rank too (quick_all_reduce.py:
    118-121, 181-200)."""
    from sglang_amd.parallel import QuickAllReduce, TPGroup

    monkeypatch.delenv("ROCM_QUICK_REDUCE_QUANTIZATION", raising=False)
    q = QuickAllReduce(None, "cpu")
    assert q.disabled and not q.size_ok(torch.float16, 1 << 22)
    tp = TPGroup(None, quick_ar=q)
    assert tp.quick_ar is None


def test_op_surface_matches_the_reference_ops_module():
    """sglang_amd.quick_ar_ops carries the names and parameter lists of the reference's quick all-reduce ops
    (custom_all_reduce_ops.py:131-163), so the reference's QuickAllReduce runs on it with `ops` swapped.  Compared against the
    reference's source where it is present (the build container); the names alone otherwise."""
    import ast
    import inspect

    import sglang_amd.quick_ar_ops as ops

    names = ["init_custom_qr", "qr_get_handle", "qr_open_handles", "qr_all_reduce", "qr_destroy", "qr_max_size"]
    for n in names:
        assert callable(getattr(ops, n))
    assert ops.IS_QUICK_AR_AVAILABLE is True
    ref = "/root/reference/python/sglang/srt/distributed/device_communicators/custom_all_reduce_ops.py"
    if not os.path.exists(ref):
        pytest.skip("reference sources not present")
    found = {}
    for node in ast.walk(ast.parse(open(ref).read())):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = [a.arg for a in node.args.args]
    assert set(found) == set(names)
    for n in names:
        mine = list(inspect.signature(getattr(ops, n)).parameters)
        assert mine == found[n], (n, mine, found[n])
