"""ctypes wrapper of oracle/librx_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/ and bench.py's cpu_baseline leg, never by sglang_amd."""
import ctypes as C
import os

import numpy as np

from . import build_oracle

_lib = None


def load():
    global _lib
    if _lib is None:
        path = build_oracle.LIB
        if not os.path.exists(path):
            build_oracle.build()
        _lib = C.CDLL(path)
        _lib.rxo_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def decode_bf16(q, k_buf, v_buf, req_to_token, req_pool_indices, seq_lens, sm_scale, logit_cap=0.0):
    """q/k_buf/v_buf: uint16 bf16 bit arrays ([bs,Hq,D], [slots,Hkv,D]); returns o bits [bs,Hq,D]."""
    q = np.ascontiguousarray(q); k_buf = np.ascontiguousarray(k_buf); v_buf = np.ascontiguousarray(v_buf)
    r2t = np.ascontiguousarray(req_to_token, dtype=np.int32)
    rpi = np.ascontiguousarray(req_pool_indices, dtype=np.int64)
    sl = np.ascontiguousarray(seq_lens, dtype=np.int64)
    bs, hq, d = q.shape
    hkv = k_buf.shape[1]
    o = np.zeros_like(q)
    rc = load().rxo_decode_bf16(_p(q), _p(k_buf), _p(v_buf), _p(o), _p(r2t), C.c_int64(r2t.shape[1]),
                                _p(rpi), _p(sl), bs, hq, hkv, d, C.c_float(sm_scale),
                                C.c_float(logit_cap))
    assert rc == 0
    return o


def extend_bf16(q, k_ext, v_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, sm_scale, causal=True):
    q = np.ascontiguousarray(q); k_ext = np.ascontiguousarray(k_ext); v_ext = np.ascontiguousarray(v_ext)
    k_buf = np.ascontiguousarray(k_buf); v_buf = np.ascontiguousarray(v_buf)
    qo = np.ascontiguousarray(qo_indptr, dtype=np.int64)
    kp = np.ascontiguousarray(kv_indptr, dtype=np.int32)
    ki = np.ascontiguousarray(kv_indices, dtype=np.int64)
    t, hq, d = q.shape
    hkv = k_ext.shape[1]
    o = np.zeros_like(q)
    rc = load().rxo_extend_bf16(_p(q), _p(k_ext), _p(v_ext), _p(o), _p(k_buf), _p(v_buf), _p(qo), _p(kp),
                                _p(ki), len(qo) - 1, hq, hkv, d, C.c_float(sm_scale), int(causal))
    assert rc == 0
    return o


def num_threads() -> int:
    return int(load().rxo_num_threads())
