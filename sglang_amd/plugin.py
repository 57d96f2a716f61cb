"""SGLang plug-in entry: registers the HIP RadixAttention backend under the name ``hip_radix``.

Wire-up on the reference side (nothing in SGLang itself changes):

  * setuptools entry point group ``sglang.srt.plugins`` -> ``sglang_amd.plugin:register``;
    SGLang's loader calls it early in every process (srt/plugins/__init__.py:29,103-141);
  * ``register`` adds the CLI choice (``add_attention_backend_choices``, srt/server_args.py:386-387)
    and the factory (``register_attention_backend``, srt/layers/attention/attention_registry.py:34-39);
  * ``--attention-backend hip_radix`` then makes ``ATTENTION_BACKENDS["hip_radix"](model_runner)``
    build this backend (model_runner_components/attention_backend_setup.py:240-246).

``sglang`` is imported lazily: the standalone harness, the tests and bench.py never need it.
"""
from __future__ import annotations

import os

BACKEND_NAME = "hip_radix"


def backend_options() -> dict:
    """Backend knobs SGLang's ServerArgs has no field for, read from the environment:
    SGLANG_HIP_RADIX_INDEX_MODE = paged | indices, SGLANG_HIP_RADIX_SPLIT_POLICY = native | reference,
    SGLANG_HIP_RADIX_CASCADE = 1 (shared-prefix decode) with
    SGLANG_HIP_RADIX_CASCADE_MIN_BS / _MIN_SHARED."""
    env = os.environ
    opts = {"decode_index_mode": env.get("SGLANG_HIP_RADIX_INDEX_MODE", "paged"),
            "split_policy": env.get("SGLANG_HIP_RADIX_SPLIT_POLICY", "native"),
            "cascade_decode": env.get("SGLANG_HIP_RADIX_CASCADE", "0") not in ("", "0", "false", "False")}
    if "SGLANG_HIP_RADIX_CASCADE_MIN_BS" in env:
        opts["cascade_min_bs"] = int(env["SGLANG_HIP_RADIX_CASCADE_MIN_BS"])
    if "SGLANG_HIP_RADIX_CASCADE_MIN_SHARED" in env:
        opts["cascade_min_shared"] = int(env["SGLANG_HIP_RADIX_CASCADE_MIN_SHARED"])
    return opts


def make_backend(model_runner):
    """Factory with the signature ATTENTION_BACKENDS expects: fn(model_runner) -> backend."""
    from .attention.backend import HipRadixAttnBackend

    return HipRadixAttnBackend(model_runner, **backend_options())


def make_sglang_backend_class():
    """Returns a subclass of SGLang's AttentionBackend ABC that delegates to HipRadixAttnBackend,
    so isinstance checks inside SGLang (hybrid wrappers, TBO) hold."""
    from sglang.srt.layers.attention.base_attn_backend import AttentionBackend

    from .attention.backend import HipRadixAttnBackend

    class SGLangHipRadixAttnBackend(HipRadixAttnBackend, AttentionBackend):
        def __init__(self, model_runner):
            AttentionBackend.__init__(self)
            HipRadixAttnBackend.__init__(self, _RunnerView(model_runner), **backend_options())

    return SGLangHipRadixAttnBackend


class _RunnerView:
    """Adapts SGLang's ModelRunner to the handful of attributes the backend reads
    (triton_backend.py:121-302 reads the same ones)."""

    def __init__(self, mr):
        self.device = mr.device
        self.req_to_token_pool = mr.req_to_token_pool
        self.token_to_kv_pool = mr.token_to_kv_pool
        self.token_to_kv_pool_allocator = getattr(mr, "token_to_kv_pool_allocator", None)
        self.page_size = getattr(mr, "page_size", 1)
        self.server_args = mr.server_args
        self.tp_size = getattr(mr, "tp_size", 1)
        # hybrid sliding-window models: window metadata is built next to the full one (triton_backend.py:259-276)
        self.sliding_window_size = getattr(mr, "sliding_window_size", None)
        self.dtype = getattr(mr, "dtype", None)  # q / o dtype (an fp8 pool cannot tell)

        class _MC:
            num_attention_heads = mr.model_config.num_attention_heads
            num_key_value_heads = mr.model_config.get_total_num_kv_heads() if hasattr(
                mr.model_config, "get_total_num_kv_heads") else mr.model_config.num_key_value_heads
            context_len = mr.model_config.context_len

        self.model_config = _MC


def register() -> None:
    """Entry-point callable (group ``sglang.srt.plugins``)."""
    from sglang.srt.layers.attention.attention_registry import register_attention_backend
    from sglang.srt.server_args import add_attention_backend_choices

    add_attention_backend_choices([BACKEND_NAME])

    @register_attention_backend(BACKEND_NAME)
    def _create(runner):
        return make_sglang_backend_class()(runner)
