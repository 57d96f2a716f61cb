"""The backend's write location against the REFERENCE's own ``KVWriteLoc`` / ``unwrap_write_loc``.

Needs /root/reference (this container only); run in its own process by tests/test_foreign_pool.py.

The reference's pools unwrap the location handed to ``set_kv_buffer`` with ``isinstance(loc_info, KVWriteLoc)``
against the dataclass defined next to them (srt/mem_cache/memory_pool.py:1531-1570, used by
MHATokenToKVPool.set_kv_buffer :2305-2316; the Triton backend builds that class at
triton_backend.py:1287-1293,1750-1753).  An instance of a DIFFERENT class named KVWriteLoc (e.g. ours) falls
through as a "bare loc" and the store receives a dataclass where it wants a tensor.  Here the two definitions
are cut out of the reference file with ``ast`` (memory_pool.py itself is not importable in this container, SURVEY
8c), executed in a fresh module together with a pool class that calls them exactly as :2305-2316 does, and the
backend is built on that pool: what it hands to ``set_kv_buffer`` must unwrap to the out_cache_loc tensor.
Nothing of the reference's text is stored: only the verdict is printed.
"""
import ast
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

REF_POOL = "/root/reference/python/sglang/srt/mem_cache/memory_pool.py"


def reference_pool_module():
    """A module holding the reference's KVWriteLoc + unwrap_write_loc and a pool shaped like its NHD
    MHATokenToKVPool (get_kv_buffer / set_kv_buffer(layer, loc_info, k, v, k_scale, v_scale))."""
    src = open(REF_POOL).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body
            if (isinstance(n, ast.ClassDef) and n.name == "KVWriteLoc")
            or (isinstance(n, ast.FunctionDef) and n.name == "unwrap_write_loc")]
    assert len(keep) == 2, [getattr(n, "name", None) for n in keep]
    mod = types.ModuleType("ref_pool_extract")
    sys.modules[mod.__name__] = mod
    pre = "from dataclasses import dataclass\nfrom typing import Optional\nimport torch\n"
    exec(compile(pre, "<pre>", "exec"), mod.__dict__)
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF_POOL, "exec"), mod.__dict__)

    import torch

    class RefShapedPool:  # memory_pool.py:2043-2094 buffers, :2305-2381 set_kv_buffer, :2295-2303 getters
        def __init__(self, size, page_size, dtype, head_num, head_dim, layer_num, device):
            self.size, self.page_size, self.dtype, self.store_dtype = size, page_size, dtype, dtype
            self.head_num, self.head_dim, self.layer_num, self.start_layer = head_num, head_dim, layer_num, 0
            self.k_buffer = [torch.zeros(size + page_size, head_num, head_dim, dtype=dtype, device=device)
                             for _ in range(layer_num)]
            self.v_buffer = [torch.zeros(size + page_size, head_num, head_dim, dtype=dtype, device=device)
                             for _ in range(layer_num)]
            self.seen = []

        def get_key_buffer(self, layer_id):
            return self.k_buffer[layer_id - self.start_layer]

        def get_value_buffer(self, layer_id):
            return self.v_buffer[layer_id - self.start_layer]

        def get_kv_buffer(self, layer_id):
            return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

        def set_kv_buffer(self, layer, loc_info, cache_k, cache_v, k_scale=None, v_scale=None,
                          layer_id_override=None, dcp_kv_mask=None):
            loc, _, _ = mod.unwrap_write_loc(loc_info)
            self.seen.append(type(loc).__name__)
            self.k_buffer[layer.layer_id][loc] = cache_k
            self.v_buffer[layer.layer_id][loc] = cache_v

    RefShapedPool.__module__ = mod.__name__
    mod.RefShapedPool = RefShapedPool
    return mod


def main():
    import _ref_import

    _ref_import.install()  # only for the two torch.cuda device queries the backend makes in __init__
    import torch

    from sglang_amd.attention import backend as be_mod
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.mem_cache import memory_pool as own

    ref = reference_pool_module()
    pool = ref.RefShapedPool(64, 16, torch.bfloat16, 2, 128, 1, "cpu")

    class MC:
        num_attention_heads, num_key_value_heads, context_len = 4, 2, 128

    runner = types.SimpleNamespace(device="cpu", tp_size=1, page_size=16, model_config=MC, sliding_window_size=None,
                                   server_args=types.SimpleNamespace(triton_attention_num_kv_splits=8,
                                                                     speculative_num_draft_tokens=None),
                                   req_to_token_pool=own.ReqToTokenPool(2, 128, "cpu"), token_to_kv_pool=pool,
                                   token_to_kv_pool_allocator=None)
    be = be_mod.HipRadixAttnBackend(runner)
    loc = torch.tensor([17, 18, 19], dtype=torch.int64)
    info = be._loc_info(loc)
    got, swa, full = ref.unwrap_write_loc(info)
    # the former behaviour, for the record: OUR dataclass does not unwrap under the reference's isinstance check
    ours_unwrapped, _, _ = ref.unwrap_write_loc(own.KVWriteLoc(loc))
    layer = RadixAttention(4, 128, 128 ** -0.5, 2, 0)
    k = torch.ones(3, 2, 128, dtype=torch.bfloat16)
    pool.set_kv_buffer(layer, info, k, k, None, None)
    res = {
        "loc_info_cls_module": type(info).__module__,
        "loc_info_is_reference_cls": type(info) is ref.KVWriteLoc,
        "unwraps_to_tensor": isinstance(got, torch.Tensor) and got.data_ptr() == loc.data_ptr(),
        "swa_full_none": swa is None and full is None,
        "own_cls_would_not_unwrap": not isinstance(ours_unwrapped, torch.Tensor),
        "pool_saw": pool.seen,
        "rows_written": bool((pool.k_buffer[0][17:20] == 1).all()) and bool((pool.k_buffer[0][:17] == 0).all()),
        "fused_store_allowed_on_foreign_pool": be._pool_allows_fused_store,
        "own_pool_resolves_own_cls": be_mod.resolve_write_loc_cls(
            own.MHATokenToKVPool(32, 16, torch.bfloat16, 2, 128, 1, "cpu")) is own.KVWriteLoc,
    }
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
