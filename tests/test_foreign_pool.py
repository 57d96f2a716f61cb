"""The backend on a KV pool that is NOT ours ("Pool methods used").

Under ``--attention-backend hip_radix`` SGLang builds its OWN MHATokenToKVPool; its ``set_kv_buffer`` unwraps the
write location with ``isinstance(loc_info, KVWriteLoc)`` against the dataclass of ITS module
(srt/mem_cache/memory_pool.py:1531-1570, :2305-2316).  Two checks:

  * container (needs /root/reference): the reference's real ``KVWriteLoc`` + ``unwrap_write_loc`` (cut out with
    ``ast``) unwrap what the backend hands over to the out_cache_loc tensor (tests/golden/check_foreign_pool.py);
  * GPU: extend + decode steps through a foreign, reference-shaped NHD pool class with its own ``KVWriteLoc``
    type, against the oracle -- every store goes through the pool's ``set_kv_buffer`` (the fused decode store is
    reserved for our pools or pools that declare ``supports_fused_decode_store``).
"""
import json
import os
import subprocess
import sys
import types
from dataclasses import dataclass
from typing import Optional

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isfile("/root/reference/python/sglang/srt/mem_cache/memory_pool.py"),
                    reason="needs the reference tree")
def test_backend_write_loc_unwraps_under_the_reference_definitions():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "check_foreign_pool.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["loc_info_is_reference_cls"] and res["unwraps_to_tensor"] and res["swa_full_none"]
    assert res["own_cls_would_not_unwrap"]  # the defect this guards against
    assert res["pool_saw"] == ["Tensor"] and res["rows_written"]
    assert not res["fused_store_allowed_on_foreign_pool"] and res["own_pool_resolves_own_cls"]


# ---------------------------------------------------------------------------------------------------------------
# a foreign pool module: its own KVWriteLoc type and unwrap (the field record of memory_pool.py:1531-1570), an NHD
# pool with the reference's method signatures (:2295-2381).  Stores are torch indexing on purpose: nothing of ours.
def _foreign_module():
    mod = types.ModuleType("foreign_kv_pool")
    sys.modules[mod.__name__] = mod

    @dataclass
    class KVWriteLoc:
        loc: torch.Tensor
        swa_loc: Optional[torch.Tensor] = None
        full_loc: Optional[torch.Tensor] = None

    def unwrap_write_loc(loc_info):
        if isinstance(loc_info, KVWriteLoc):
            return loc_info.loc, loc_info.swa_loc, loc_info.full_loc
        return loc_info, None, None

    class ForeignMHAPool:
        def __init__(self, size, page_size, dtype, head_num, head_dim, layer_num, device):
            self.size, self.page_size, self.dtype, self.store_dtype = size, page_size, dtype, dtype
            self.head_num, self.head_dim, self.layer_num, self.start_layer = head_num, head_dim, layer_num, 0
            self.device = device
            mk = lambda: [torch.zeros(size + page_size, head_num, head_dim, dtype=dtype, device=device)  # noqa: E731
                          for _ in range(layer_num)]
            self.k_buffer, self.v_buffer = mk(), mk()
            self.calls = 0       # the "side effect" a fused store would skip
            self.loc_types = set()

        def get_key_buffer(self, layer_id):
            return self.k_buffer[layer_id - self.start_layer]

        def get_value_buffer(self, layer_id):
            return self.v_buffer[layer_id - self.start_layer]

        def get_kv_buffer(self, layer_id):
            return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

        def set_kv_buffer(self, layer, loc_info, cache_k, cache_v, k_scale=None, v_scale=None,
                          layer_id_override=None, dcp_kv_mask=None):
            loc, _, _ = unwrap_write_loc(loc_info)
            self.loc_types.add(type(loc_info).__module__ + "." + type(loc_info).__name__)
            if not isinstance(loc, torch.Tensor):
                raise TypeError(f"set_kv_buffer: loc is {type(loc)} (a KVWriteLoc of another module?)")
            self.calls += 1
            lid = layer_id_override if layer_id_override is not None else layer.layer_id
            self.k_buffer[lid - self.start_layer][loc.long()] = cache_k.view(-1, self.head_num, self.head_dim)
            self.v_buffer[lid - self.start_layer][loc.long()] = cache_v.view(-1, self.head_num, self.head_dim)

    for o in (KVWriteLoc, unwrap_write_loc, ForeignMHAPool):
        o.__module__ = mod.__name__
        setattr(mod, o.__name__, o)
    return mod


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _f64(t):  # a kernel OUTPUT as float64 values (not bit patterns)
    return t.detach().float().cpu().numpy().astype(np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("page_size", [1, 16])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_extend_and_decode_through_a_foreign_reference_shaped_pool(page_size, dtype):
    import parity_util as parity
    from oracle import radix_oracle as orc
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache import memory_pool as own

    DEV = "cuda"
    fm = _foreign_module()
    hq, hkv, d, ps = 8, 2, 128, page_size
    size, max_ctx = 4096, 1024
    pool = fm.ForeignMHAPool(size, ps, dtype, hkv, d, 1, DEV)
    r2t = own.ReqToTokenPool(4, max_ctx, DEV)

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, hkv, max_ctx

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        token_to_kv_pool_allocator = None
        model_config = MC

        class server_args:
            triton_attention_num_kv_splits = 8

    MR.page_size = ps
    be = HipRadixAttnBackend(MR)
    assert be._write_loc_cls is fm.KVWriteLoc and not be._pool_allows_fused_store
    layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
    g = torch.Generator().manual_seed(3)
    rand = lambda *s: torch.randn(*s, generator=g).to(dtype).to(DEV)  # noqa: E731

    # slots: whole pages handed out in a shuffled order (page 0 = the padding page stays unused)
    prefix, extend = [70, 0, 257], [33, 64, 5]
    bs = len(prefix)
    rows = r2t.alloc(bs)
    perm = (torch.randperm(size // ps - 1, generator=g) + 1).tolist()
    row_slots = []
    for i in range(bs):
        need = -(-(prefix[i] + extend[i] + 4) // ps)
        pages, perm = perm[:need], perm[need:]
        sl = torch.tensor([p * ps + j for p in pages for j in range(ps)], dtype=torch.int32)
        row_slots.append(sl)
        r2t.req_to_token[rows[i], : len(sl)] = sl.to(DEV)
    # cached prefix: written through the pool's own API (as SGLang's earlier forwards would have)
    for i in range(bs):
        if prefix[i]:
            loc = row_slots[i][: prefix[i]].to(torch.int64).to(DEV)
            pool.set_kv_buffer(layer, fm.KVWriteLoc(loc), rand(prefix[i], hkv, d), rand(prefix[i], hkv, d))
    calls0 = pool.calls
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)

    # ---- extend over the cached prefixes
    seq = [p + e for p, e in zip(prefix, extend)]
    loc = torch.cat([row_slots[i][prefix[i]: seq[i]] for i in range(bs)]).to(torch.int64).to(DEV)
    T = sum(extend)
    q, k, v = rand(T, hq * d), rand(T, hkv * d), rand(T, hkv * d)
    fb = ForwardBatch.for_extend(rpi, torch.tensor(seq, device=DEV), loc, list(prefix), list(extend))
    be.init_forward_metadata(fb)
    o = layer(q, k, v, fb, be)
    torch.cuda.synchronize()
    assert pool.calls == calls0 + 1
    kb, vb = pool.get_kv_buffer(0)
    want = orc.sdpa_extend_req_to_token(_bits(q.view(T, hq, d)), _bits(kb), _bits(vb), _bits(r2t.req_to_token),
                                        np.array(rows), np.array(seq), np.array(prefix), np.array(extend), d ** -0.5)
    absw = None
    if dtype == torch.bfloat16:  # bf16 P rounding on the first causal rows (parity_util.check_out)
        absw = orc.sdpa_extend_req_to_token(_bits(q.view(T, hq, d)), _bits(kb), parity.abs_values(_bits(vb)),
                                            _bits(r2t.req_to_token), np.array(rows), np.array(seq), np.array(prefix),
                                            np.array(extend), d ** -0.5)
    parity.check_out(_f64(o.view(T, hq, d)), want, dtype, ("foreign_pool_extend", ps), absw=absw)

    # ---- two decode steps: each store must go through the pool's set_kv_buffer (no fused store on a foreign pool)
    for step in range(2):
        seq = [s + 1 for s in seq]
        loc = torch.stack([row_slots[i][seq[i] - 1] for i in range(bs)]).to(torch.int64).to(DEV)
        seq_t = torch.tensor(seq, dtype=torch.int64)
        q, k, v = rand(bs, hq * d), rand(bs, hkv * d), rand(bs, hkv * d)
        fb = ForwardBatch.for_decode(rpi, seq_t.to(DEV), loc, seq_t)
        be.init_forward_metadata(fb)
        o = layer(q, k, v, fb, be)
        torch.cuda.synchronize()
        assert pool.calls == calls0 + 2 + step
        kb, vb = pool.get_kv_buffer(0)
        # the new rows are where out_cache_loc says, with the values handed in
        assert torch.equal(kb[loc], k.view(bs, hkv, d)) and torch.equal(vb[loc], v.view(bs, hkv, d))
        want = orc.sdpa_decode_req_to_token(_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb), _bits(r2t.req_to_token),
                                            np.array(rows), np.array(seq), d ** -0.5)
        parity.check_out(_f64(o.view(bs, hq, d)), want, dtype, ("foreign_pool_decode", ps, step))
    assert pool.loc_types == {"foreign_kv_pool.KVWriteLoc"}

    # a foreign pool that opts in gets the fused store (and then its set_kv_buffer is NOT called on decode)
    pool.supports_fused_decode_store = True
    be2 = HipRadixAttnBackend(MR)
    assert be2._pool_allows_fused_store
    seq = [s + 1 for s in seq]
    loc = torch.stack([row_slots[i][seq[i] - 1] for i in range(bs)]).to(torch.int64).to(DEV)
    seq_t = torch.tensor(seq, dtype=torch.int64)
    q, k, v = rand(bs, hq * d), rand(bs, hkv * d), rand(bs, hkv * d)
    fb = ForwardBatch.for_decode(rpi, seq_t.to(DEV), loc, seq_t)
    be2.init_forward_metadata(fb)
    before = pool.calls
    o = layer(q, k, v, fb, be2)
    torch.cuda.synchronize()
    assert pool.calls == before
    kb, vb = pool.get_kv_buffer(0)
    assert torch.equal(kb[loc], k.view(bs, hkv, d)) and torch.equal(vb[loc], v.view(bs, hkv, d))
    want = orc.sdpa_decode_req_to_token(_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb), _bits(r2t.req_to_token),
                                        np.array(rows), np.array(seq), d ** -0.5)
    parity.check_out(_f64(o.view(bs, hq, d)), want, dtype, ("foreign_pool_decode_fused", ps))
