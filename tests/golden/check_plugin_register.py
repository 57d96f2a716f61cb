"""Executes sglang_amd.plugin.register() against the REFERENCE's own registry objects and builds the backend
through the registered factory from a reference-shaped ModelRunner stub.  Needs /root/reference (this container
only); run in its own process by tests/test_plugin_register.py because the import shims patch sys.modules and
torch.cuda queries.

What is real here: sglang.srt.layers.attention.attention_registry (ATTENTION_BACKENDS, register_attention_backend,
attention_registry.py:28-39), sglang.srt.server_args (ATTENTION_BACKEND_CHOICES, add_attention_backend_choices,
server_args.py:386-387) and the AttentionBackend ABC (base_attn_backend.py:20-281).  What is stubbed: optional
third-party packages (tests/golden/_ref_import.py), the package __init__ of sglang.srt.configs (it imports every
model config, one of which needs torchvision) and three server_args imports that build pydantic / transformers
objects at import time.
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import  # noqa: E402


def main():
    _ref_import._StubFinder.ROOTS = tuple(r for r in _ref_import._StubFinder.ROOTS if r != "xgrammar")
    _ref_import.install()
    sys.modules["msgspec.structs"] = sys.modules["msgspec"].structs

    class _Pkg(types.ModuleType):  # `from sglang.srt.configs import XConfig` -> a placeholder class
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            cls = type(name, (), {})
            setattr(self, name, cls)
            return cls

    pkg = _Pkg("sglang.srt.configs")
    pkg.__path__ = [os.path.join(_ref_import.REF_PY, "sglang/srt/configs")]
    sys.modules["sglang.srt.configs"] = pkg
    for name in ("sglang.srt.function_call.function_call_parser", "sglang.srt.parser.reasoning_parser",
                 "sglang.srt.utils.hf_transformers_utils"):
        sys.modules[name] = _ref_import._Stub(name)

    import torch
    from sglang.srt.layers.attention import attention_registry as reg
    from sglang.srt.layers.attention.base_attn_backend import AttentionBackend
    from sglang.srt import server_args as sa

    from sglang_amd import plugin
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

    before = set(reg.ATTENTION_BACKENDS)
    plugin.register()
    res = {"choice_added": plugin.BACKEND_NAME in sa.ATTENTION_BACKEND_CHOICES,
           "factory_added": plugin.BACKEND_NAME in reg.ATTENTION_BACKENDS,
           "others_kept": before <= set(reg.ATTENTION_BACKENDS)}

    # a runner with the attribute surface TritonAttnBackend.__init__ reads (triton_backend.py:121-302)
    class ModelConfig:
        num_attention_heads, context_len = 32, 4096

        def get_total_num_kv_heads(self):
            return 8

        def get_num_kv_heads(self, tp):
            return max(1, 8 // tp)

    runner = types.SimpleNamespace(
        device="cpu", gpu_id=0, tp_size=2, page_size=16, dtype=torch.bfloat16, sliding_window_size=None,
        model_config=ModelConfig(), use_mla_backend=False,
        server_args=types.SimpleNamespace(triton_attention_num_kv_splits=8, speculative_num_draft_tokens=None,
                                          attention_backend=plugin.BACKEND_NAME),
        req_to_token_pool=ReqToTokenPool(4, 4096, "cpu"),
        token_to_kv_pool=MHATokenToKVPool(256, 16, torch.bfloat16, 4, 128, 1, "cpu"),
        token_to_kv_pool_allocator=None)
    be = reg.ATTENTION_BACKENDS[plugin.BACKEND_NAME](runner)
    res.update(isinstance_abc=isinstance(be, AttentionBackend), cls=type(be).__name__,
               num_head=be.num_head, num_kv_head=be.num_kv_head, page_size=be.page_size,
               max_kv_splits=be.max_kv_splits, cu=be.device_core_count,
               abstract_left=sorted(getattr(type(be), "__abstractmethods__", ())))
    # every hook the runners call on a backend resolves to OUR implementation, not the ABC's default
    ours = []
    for name in ("init_forward_metadata", "init_forward_metadata_out_graph", "init_forward_metadata_in_graph",
                 "init_cuda_graph_state", "get_cuda_graph_seq_len_fill_value", "forward", "forward_decode",
                 "forward_extend"):
        ours.append(getattr(type(be), name).__qualname__.startswith("HipRadixAttnBackend."))
    res["hooks_are_ours"] = all(ours)
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
