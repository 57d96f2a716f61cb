"""HipRadixAttnBackend: the drop-in AttentionBackend for the RadixAttention path on MI355X.

Mirrors the method set the reference's runners call on a backend
(srt/layers/attention/base_attn_backend.py:21-281) and the behaviour of its closest
analogue TritonAttnBackend (srt/layers/attention/triton_backend.py:116-1864):

  __init__(model_runner)                     triton_backend.py:121-302
  init_forward_metadata(fb)                  :714-960   (once per eager forward)
  forward(q, k, v, layer, fb, save_kv_cache) base_attn_backend.py:188-231
  forward_decode / forward_extend            :1714-1864 / :1250-1437
  init_cuda_graph_state, init_forward_metadata_out_graph / _in_graph,
  get_cuda_graph_seq_len_fill_value          :962-1206

Every kernel is a call into libradix_hip.so (sglang_amd.ops); there is no torch or CPU fallback.

Two decode index modes:
  * "paged"   (default, MI355X-native): the decode kernel walks req_to_token itself, so a
    decode forward builds NO kv_indices (the reference rewrites 8 MiB of int64 per step at
    bs=256/ctx=4k) and, when the host-side split schedule says one pass suffices, no fp32
    partials either.
  * "indices" (reference contract): kv_indptr / kv_indices / num_kv_splits / attn_logits /
    attn_lse exactly as TritonAttnBackend.ForwardMetadata (:91-113) lays them out.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

import numpy as np
import torch

from .. import ops
from ..forward_batch import ForwardBatch
from ..mem_cache import memory_pool as _own_pools


def draft_kv_indices_buffer_width(num_seqs: int, topk: int, max_context_len: int) -> int:
    """Row width of the multi-step draft kv_indices buffer (srt/speculative/spec_utils.py:181-192)."""
    assert num_seqs * topk * max_context_len < 2 ** 31, "kv_indices flat offset would overflow int32; reduce batch/topk/context"
    return num_seqs * topk * max_context_len


def draft_kv_indices_used_len(seq_lens_sum: int, topk: int, bs: int, num_steps: int) -> int:
    """kv_indices words in use after num_steps draft steps (spec_utils.py:195-203)."""
    return seq_lens_sum * topk + bs * num_steps


def split_pairs_bound(bs: int, slots: int, wgpr: int, cus: int) -> int:
    """max over all length vectors of sum(rx_num_kv_splits_balanced(...)) for ``bs`` requests, cap ``slots``, ``wgpr``
    workgroups per (request, split) pair and wg_target = 2 x ``cus`` (any wg_target_mixed the backend passes), rule by rule:
    * even share / rounds rule / 3-per-CU mixed budget: a request takes <= ceil(len / t*) <= len / t* + 1 pieces with
      t* >= work / budget, so sum <= bs + budget / wgpr + 1, budget <= 3 x cus;
    * fill rule, near-uniform batch below 3 workgroups per CU (live x wgpr < 3 cus): ONE count <= min(slots, 6) for every
      live request (257 requests x 6 = 1542 pairs against the old bound's 1026);
    * fill rule, everybody cut (live x wgpr < 0.7 cus): one count <= ceil(2 cus / (live x wgpr)) + 8."""
    even = bs + -(-3 * cus // wgpr) + 1
    live3 = min(bs, max(0, -(-3 * cus // wgpr) - 1))         # most live requests the < 3-per-CU rule admits
    fill = (min(slots, 6) - 1) * live3 + bs
    live07 = min(bs, max(0, -(-7 * cus // (10 * wgpr))))     # ... the everybody-is-cut rule admits
    cut = -(-2 * cus // wgpr) + 9 * live07 + bs
    return min(bs * slots, max(even, fill, cut))


@dataclass
class ForwardMetadata:
    """Subset of triton_backend.py:91-113 that the dense path uses."""

    attn_logits: Optional[torch.Tensor]
    attn_lse: Optional[torch.Tensor]
    max_extend_len: Optional[int]
    num_kv_splits: Optional[torch.Tensor]
    kv_indptr: Optional[torch.Tensor]
    kv_indices: Optional[torch.Tensor]
    qo_indptr: Optional[torch.Tensor]
    max_kv_splits: int = 1
    # speculative decoding (TARGET_VERIFY / DRAFT_EXTEND_V2): the draft tree's mask
    custom_mask: Optional[torch.Tensor] = None
    mask_indptr: Optional[torch.Tensor] = None
    # sliding-window layers: the last min(len, W) tokens of every request
    window_kv_indptr: Optional[torch.Tensor] = None
    window_kv_indices: Optional[torch.Tensor] = None
    window_num_kv_splits: Optional[torch.Tensor] = None
    window_kv_offsets: Optional[torch.Tensor] = None
    # decode: launch order of the requests (longest first) for batches that take more than one round of workgroups
    request_order: Optional[torch.Tensor] = None
    # decode, length-aware schedule: upper estimate of the (request, split) pairs that write a partial
    partial_pairs_hint: int = 0
    # decode, length-aware schedule: the live (request, split) pairs, compacted (ops.SplitItems)
    split_items: Optional[object] = None
    decode_units: Optional[object] = None   # ops.DecodeUnits of this forward (req_to_token mode; rx_decode_params.unit_desc)
    draft: bool = False  # kv_indptr / kv_indices came from spec_info (multi-step draft decode: one row per top-k branch)
    # deterministic inference: the unified kv list of this forward, built by the first layer that needs it --
    # {is sliding-window layer: (unified_kv_indptr, unified_kv_indices, prefix_lens)}
    unified: Optional[dict] = None


def host_num_kv_splits(seq_lens: np.ndarray, num_head: int, num_kv_head: int, max_kv_splits: int,
                       device_core_count: int) -> np.ndarray:
    """Host restatement of get_num_kv_splits_triton (kernels/ops/attention/metadata.py:11-60)
    used only to pick the launch shape (single pass vs split-KV) without a device sync; the
    values the kernels consume come from rx_num_kv_splits."""
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    num_seq = len(seq_lens)
    mx, mn = int(seq_lens.max()), int(seq_lens.min())
    if mx * 8 < mn * 10:
        mn = mx
    mn = max(mn, 1)
    s1 = min(-(-mx // mn), max_kv_splits)
    c1 = -(-mx // s1)
    ext = np.float32(mx) / np.float32(64.0)
    cores = int(np.float32(device_core_count) * np.maximum(np.log2(ext, dtype=np.float32), np.float32(1.0)))
    group = num_head // num_kv_head
    if group == 1:
        token_grid = num_seq * num_head
    else:
        token_grid = num_seq * (-(-num_head // min(16, group)))
    s2 = max(1, min(-(-cores // token_grid), max_kv_splits))
    c2 = -(-mx // s2)
    return np.maximum(-(-seq_lens // c1), -(-seq_lens // c2)).astype(np.int32)


def resolve_write_loc_cls(pool):
    """The KVWriteLoc class ``pool.set_kv_buffer`` unwraps -- the one defined next to the POOL, never ours for a
    foreign pool.  The reference's pools test ``isinstance(loc_info, KVWriteLoc)`` against THEIR dataclass
    (unwrap_write_loc, srt/mem_cache/memory_pool.py:1566-1570; built by the Triton backend at
    triton_backend.py:1287-1293,1750-1753): an instance of another class of the same name falls through as the
    "bare loc" and the store dies on a dataclass where it wants a tensor.  Order: our own pools -> our class; a
    foreign pool -> ``KVWriteLoc`` of the module (or a base class's module) that defines the pool; then of
    sglang's memory_pool module if that is ALREADY imported (never imported from here); else None = hand the
    bare ``out_cache_loc`` tensor over, which unwrap_write_loc accepts as it stands."""
    import sys

    if isinstance(pool, (_own_pools.MHATokenToKVPool, _own_pools.MLATokenToKVPool)):
        return _own_pools.KVWriteLoc
    for cls in type(pool).__mro__:
        mod = sys.modules.get(getattr(cls, "__module__", None) or "")
        cand = getattr(mod, "KVWriteLoc", None)
        if isinstance(cand, type) and cand is not _own_pools.KVWriteLoc:
            return cand
    mod = sys.modules.get("sglang.srt.mem_cache.memory_pool")
    cand = getattr(mod, "KVWriteLoc", None)
    return cand if isinstance(cand, type) else None


def md_has_mask(md) -> bool:
    return md is not None and getattr(md, "custom_mask", None) is not None


class HipRadixAttnBackend:
    needs_cpu_seq_lens: bool = True
    supports_ragged_verify_graph: bool = False

    def __init__(self, model_runner, decode_index_mode: str = "paged",
                 max_kv_splits: Optional[int] = None, split_policy: str = "native",
                 cascade_decode: bool = False, cascade_min_bs: int = 16, cascade_min_shared: int = 1024,
                 dcp=None, mla_v_is_latent_prefix: bool = False, skip_prefill: bool = False,
                 kv_indptr_buf: Optional[torch.Tensor] = None, topk: int = 1):
        self.device = model_runner.device
        self.topk = max(1, int(topk))  # draft backends: rows per request (one per top-k branch)
        self.skip_prefill = bool(skip_prefill)      # (signature parity with TritonAttnBackend: a draft-decode-only backend)
        self._kv_indptr_buf = kv_indptr_buf
        # MLA pools keep ONE latent row per token and serve v as its first kv_lora_rank columns (the reference's
        # MLATokenToKVPool.set_kv_buffer drops cache_v); the model hands forward_extend v = k_nope, a tensor of its own
        # with the same values.  With this flag the extend reads the new tokens' v from their k rows as well -- the
        # kernel's one-image form -- instead of from v.  Off by default: it is the caller's statement that
        # v == k[..., :kv_lora_rank] for the new tokens, which the absorbed formulation guarantees and the API does not.
        self.mla_v_is_latent_prefix = bool(mla_v_is_latent_prefix)
        self.req_to_token_pool = model_runner.req_to_token_pool
        self.token_to_kv_pool = model_runner.token_to_kv_pool
        self.token_to_kv_pool_allocator = getattr(model_runner, "token_to_kv_pool_allocator", None)
        self.req_to_token = self.req_to_token_pool.req_to_token
        self.page_size = getattr(model_runner, "page_size", 1) or 1
        mc = model_runner.model_config
        tp = getattr(model_runner, "tp_size", 1)
        self.num_head = mc.num_attention_heads // tp
        self.num_kv_head = max(1, mc.num_key_value_heads // tp)
        # decode context parallel (attention/dcp.py): a request's KV is spread over the group's ranks; every rank runs
        # the group's gathered q heads over its own tokens (triton_backend.py:182-190: num_head is the gathered count)
        self.dcp = dcp if (dcp is not None and dcp.size > 1) else None
        self.local_num_head = self.num_head
        if self.dcp is not None:
            # DCP spreads the tokens over the TP ranks that would otherwise hold COPIES of one kv head (tp_size > kv
            # heads): the gathered q heads are that head's GQA group.  With several kv heads per rank the GQA mapping
            # of the local heads (new tokens' block) and of the gathered heads (cached part) would disagree.
            if self.num_kv_head != 1:
                raise ValueError(f"DCP needs one kv head per rank (the replicated-kv-head case), got {self.num_kv_head}")
            self.num_head *= self.dcp.size
            decode_index_mode = "indices"  # the rank's share of a request is a strided subset of its row
        self.v_head_dim = self.token_to_kv_pool.get_value_buffer(
            getattr(self.token_to_kv_pool, "start_layer", 0)).shape[-1]
        self.max_context_len = mc.context_len
        sa = getattr(model_runner, "server_args", None)
        self.max_kv_splits = max_kv_splits or getattr(sa, "triton_attention_num_kv_splits", 8)
        self.device_core_count = torch.cuda.get_device_properties(self.device).multi_processor_count
        # "native": the MI355X schedule (one to two workgroups per CU in total, rx_num_kv_splits_native;
        # up to native_split_cap splits) -- "reference": get_num_kv_splits_triton's formula (K3) with
        # --triton-attention-num-kv-splits as the cap
        if split_policy not in ("native", "reference"):
            raise ValueError(f"split_policy must be 'native' or 'reference', got {split_policy}")
        # --enable-deterministic-inference (triton_backend.py:247-264, :325-333, :1339-1350): results that depend on the
        # REQUEST only, not on the batch it rides in nor on where the radix cache cut its prompt.  Decode: a request is
        # cut into ceil(len / tile) splits of a fixed tile (SGLANG_TRITON_DECODE_SPLIT_TILE_SIZE, 256) whatever else the
        # batch holds -- no chip-filling schedule, no shared-prefix cascade.  Extend: the ONE-stage kernel over the unified
        # kv list (prefix slots + the new tokens' slots, rx_build_unified_kv_indices), whose tiles are cut from the list's
        # start and all run one tile body (rx_extend32_kernel.inc), so a query row's arithmetic does not move with the
        # prefix / extend split.
        # (:235-245) safe only when a prefill is never split or padded: no chunked prefill, and no graph mode that could
        # (the runner says so through server_args.disable_cuda_graph; eager prefill is what this backend runs)
        self.allow_bidirectional_attention_in_extend = (getattr(sa, "chunked_prefill_size", None) == -1
                                                        and bool(getattr(sa, "disable_cuda_graph", True)))
        self.static_kv_splits = os.environ.get("SGLANG_TRITON_DECODE_ATTN_STATIC_KV_SPLITS", "false").lower() in ("1", "true")
        if self.static_kv_splits:
            split_policy = "reference"  # (every request gets the cap: the reference's K3 path, not the chip-filling one)
        self.enable_deterministic = bool(getattr(sa, "enable_deterministic_inference", False))
        self.split_tile_size = getattr(sa, "triton_attention_split_tile_size", None)
        if self.enable_deterministic:
            self.split_tile_size = int(os.environ.get("SGLANG_TRITON_DECODE_SPLIT_TILE_SIZE", "256"))
            split_policy, cascade_decode = "reference", False
        if self.split_tile_size is not None:
            self.max_kv_splits = (self.max_context_len + self.split_tile_size - 1) // self.split_tile_size
        self.split_policy = split_policy
        self.native_split_cap = 32
        # graph replay: upper bound of the fp32 kv-split partials [bs, Hq, slots, Dv] kept address-stable (bytes)
        self.graph_partials_budget = int(getattr(sa, "rx_graph_partials_budget", 256 << 20) or (256 << 20))
        # workgroups the native schedule aims for: one per CU for the dense kernel (four independent waves
        # each); two per CU for the MLA kernel, whose four waves share one staged tile (config-5 shape:
        # 256 workgroups 157 us, 512 workgroups 134 us)
        self._is_mla_pool = hasattr(self.token_to_kv_pool, "kv_lora_rank")
        # the write-location wrapper of THIS pool's module (a foreign pool unwraps its own class only) and whether
        # the decode kernel may write the step's new K/V row itself: our pools, or a pool that says so -- a foreign
        # set_kv_buffer may carry side effects the fused store would skip (layer-transfer sync memory_pool.py:
        # 2273-2279, the OOB probe :2319, canaries)
        self._own_pool = isinstance(self.token_to_kv_pool, (_own_pools.MHATokenToKVPool, _own_pools.MLATokenToKVPool))
        self._write_loc_cls = resolve_write_loc_cls(self.token_to_kv_pool)
        self._pool_allows_fused_store = self._own_pool or bool(
            getattr(self.token_to_kv_pool, "supports_fused_decode_store", False))
        if decode_index_mode not in ("paged", "indices"):
            raise ValueError(f"decode_index_mode must be 'paged' or 'indices', got {decode_index_mode}")
        self.decode_index_mode = decode_index_mode
        max_bs = self.req_to_token_pool.size
        self.kv_indptr = (self._kv_indptr_buf if self._kv_indptr_buf is not None
                          else torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device))
        self.qo_indptr = torch.zeros((max_bs + 1,), dtype=torch.int64, device=self.device)
        self.mask_indptr = torch.zeros((max_bs + 1,), dtype=torch.int64, device=self.device)
        # hybrid sliding-window models (triton_backend.py:259-276): a second set of window indices
        self.sliding_window_size = getattr(model_runner, "sliding_window_size", None)
        if self.sliding_window_size is not None and self.sliding_window_size > 0:
            self.window_kv_indptr = torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device)
        else:
            self.sliding_window_size, self.window_kv_indptr = None, None
        self.num_draft_tokens = getattr(sa, "speculative_num_draft_tokens", None)
        self.forward_metadata: Optional[ForwardMetadata] = None
        self._scratch_logits = None
        self._scratch_lse = None
        # stage 2 inside the stage-1 kernel (rx_decode_params.merge_counters): one zeroed word per (request, head);
        # the kernels leave it zero, so one buffer serves every layer and every replay of a captured step
        self._merge_counters = torch.zeros(max(1, self.req_to_token_pool.size) * self.num_head * self.topk,
                                           dtype=torch.int32, device=self.device)  # (draft backends: topk branch rows per request)
        if os.environ.get("RX_NO_INKERNEL_MERGE"):  # dev A/B: stage 2 as its own launch
            self._merge_counters = None
        self._no_fused_store = bool(os.environ.get("RX_NO_FUSED_STORE"))  # dev A/B: the store as its own launch
        self._split_occ3 = os.environ.get("RX_SPLIT_OCC3", "0") == "1"  # eager mixed batches on the three-per-CU kernel form
        self._no_decode_units = bool(os.environ.get("RX_NO_DECODE_UNITS"))  # dev A/B: the prologue's own chain of loads
        self._no_split_items = bool(os.environ.get("RX_NO_SPLIT_ITEMS"))  # dev A/B: split slots instead of compacted pairs
        # RX_DEBUG_CHECKS=1: host-synchronising assertions of the backend's preconditions (see forward_decode)
        self._debug_checks = os.environ.get("RX_DEBUG_CHECKS", "0") not in ("", "0")
        self._roctx = False
        self._graph = None  # static buffers of init_cuda_graph_state
        self._md_version = 0  # bumped by every init_forward_metadata_out_graph
        self._decode_launchers = {}  # layer_id -> ops.DecodeLauncher
        self._cur_fb = None
        # shared-prefix (cascade) decode, SURVEY 8f-2: opt-in (a batch without a common prefix pays two empty
        # launches per layer); the common prefix itself is found on the device every forward
        if self.dcp is not None and self.sliding_window_size is not None:
            raise NotImplementedError("DCP here covers pools without sliding-window layers")
        self.cascade_decode = bool(cascade_decode) and self.sliding_window_size is None and self.dcp is None
        self.cascade_min_bs, self.cascade_min_shared = int(cascade_min_bs), int(cascade_min_shared)
        self._cascade = None
        self._cascade_on = False
        self._groups_plan_cache = None    # (node ids, groups) of the last tree plan
        self._cascade_groups = None       # ops.CascadeGroups: several prefixes (forward_batch.radix_last_nodes / cascade_groups)
        self._cascade_call = None
        self.cascade_min_members = 4      # a group of fewer requests re-reads its prefix instead
        self.cascade_groups_max_shared = 1 << 20  # slots of all shared prefixes of one batch together
        self._verify_split = None      # ops.VerifySplitKV, built on the first TARGET_VERIFY forward
        self._verify_split_on = False
        self._extend_split_on = False
        self._model_dtype = getattr(model_runner, "dtype", None)

    def _split_dims(self):
        """(Dk, Dv) the split-KV extend forms serve for this pool, or None: head dim 128, or 16-bit latent MLA rows
        (576 / 512 over one kv head)."""
        if self._is_mla_pool:
            pool = self.token_to_kv_pool
            kb = pool.get_key_buffer(getattr(pool, "start_layer", 0))
            if kb.dtype in (torch.bfloat16, torch.float16) and kb.shape[-1] == 576 and self.v_head_dim == 512:
                return 576, 512
            return None
        return (128, 128) if self.v_head_dim == 128 else None

    def _q_dtype(self, k_buffer: torch.Tensor):
        """dtype of q / o: the runner's model dtype, else the pool's if that is a 16-bit float."""
        dt = getattr(self, "_model_dtype", None)
        if dt in (torch.bfloat16, torch.float16):
            return dt
        return k_buffer.dtype if k_buffer.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16

    # ------------------------------------------------------------------ scratch
    def _scratch(self, bs: int, splits: Optional[int] = None):
        """fp32 partials [bs, Hq, S, Dv] / [bs, Hq, S] carved out of flat, grow-only buffers."""
        S = self.max_kv_splits if splits is None else splits
        n = bs * self.num_head * S
        if self._scratch_lse is None or self._scratch_lse.numel() < n:
            self._scratch_logits = torch.empty(n * self.v_head_dim, dtype=torch.float32, device=self.device)
            self._scratch_lse = torch.empty(n, dtype=torch.float32, device=self.device)
        return (self._scratch_logits[: n * self.v_head_dim].view(bs, self.num_head, S, self.v_head_dim),
                self._scratch_lse[:n].view(bs, self.num_head, S))

    # ------------------------------------------------------------------ metadata
    def init_forward_metadata(self, forward_batch: ForwardBatch):
        """Eager entry point (triton_backend.py:714-960): per-call tensors, host-side launch-shape decisions."""
        self._build_metadata(forward_batch, graph=False)

    def init_forward_metadata_in_graph(self, forward_batch: ForwardBatch):
        """Graph-recordable part: nothing -- the kernels read seq_lens / req_to_token directly."""

    def init_forward_metadata_out_graph(self, forward_batch: ForwardBatch, in_capture: bool = False):
        """Graph entry point (triton_backend.py:572-632), called by the runner before capture
        (``in_capture=True``) AND before every ``graph.replay()`` (decode_cuda_graph_runner.py:1168): it
        always (re)fills the address-stable buffers of ``init_cuda_graph_state`` -- num_kv_splits, the fp32
        partials, kv_indices / kv_indptr, the verify qo / mask indptr -- because those are what the captured
        kernels read.  Every launch-shape decision here depends on ``bs`` only, never on the lengths.  A
        runner that never built graph state gets the eager body (base_attn_backend.py:55-56)."""
        self._build_metadata(forward_batch, graph=self._graph is not None)
        if self._graph is not None:
            # from here on a captured graph may hold the static buffers' addresses: nothing may be (re)allocated
            self._graph["captured"] = True

    def _build_metadata(self, forward_batch: ForwardBatch, graph: bool):
        from .. import lib as _lib
        self._roctx = bool(_lib.get_option("roctx"))
        bs = forward_batch.batch_size
        mode = forward_batch.forward_mode
        self._md_version += 1
        if mode.is_idle():
            self.forward_metadata = ForwardMetadata(None, None, None, None, None, None, None)
            return
        draft = mode.is_decode() and forward_batch.spec_info is not None and getattr(forward_batch.spec_info, "kv_indptr", None) is not None
        if graph and not draft and bs > self._graph["max_bs"]:
            raise ValueError(f"batch size {bs} exceeds init_cuda_graph_state's max_bs {self._graph['max_bs']}")
        if self.dcp is not None:
            if mode.is_decode():
                self.forward_metadata = self._decode_metadata_dcp(forward_batch, bs, graph)
            elif mode.is_target_verify() or mode.is_draft_extend_v2():
                raise NotImplementedError("speculative modes under DCP")
            else:
                self.forward_metadata = self._extend_metadata_dcp(forward_batch, bs)
            return
        if mode.is_decode():
            self.forward_metadata = self._decode_metadata(forward_batch, bs, graph)
        elif mode.is_target_verify():
            self.forward_metadata = self._target_verify_metadata(forward_batch, bs, graph)
        elif mode.is_draft_extend_v2():
            self.forward_metadata = self._draft_extend_metadata(forward_batch, bs)
        else:
            self.forward_metadata = self._extend_metadata(forward_batch, bs)

    # ------------------------------------------------------------------ sliding window
    def _window(self, lens: torch.Tensor, req_pool_indices: torch.Tensor, bs: int, graph: bool = False):
        """update_sliding_window_buffer (triton_backend.py:2043-2110): the last min(len, W) slots of every
        request -> (window_kv_indptr, window_kv_indices, window_kv_lens, window_kv_offsets)."""
        w = self.sliding_window_size
        window_lens = torch.clamp(lens, max=w)
        start = (lens - window_lens).to(torch.int32)
        total = min(bs * w, bs * self.max_context_len)
        if graph:
            kv_indices = self._graph["window_kv_indices"]
            self._graph["window_kv_offsets"][:bs].copy_(start)
            start = self._graph["window_kv_offsets"][:bs]
        else:
            kv_indices = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
        kv_indptr = ops.build_kv_indices(self.req_to_token, req_pool_indices, window_lens,
                                         self.window_kv_indptr, kv_indices, start)
        return kv_indptr, kv_indices, window_lens, start

    def _request_order(self, fb: ForwardBatch, bs: int, use_graph_bufs: bool, force: bool = False) -> Optional[torch.Tensor]:
        """rx_decode_params.request_order: the batch's requests by descending length, when the launch has more
        workgroups than the chip holds at once (two per CU) -- a ragged batch's longest requests then start first and
        its last round is the short ones (bs 256, lengths uniform in [2 k, 4 k]: decode kernel 0.73 -> 0.76 of HBM peak)."""
        group = max(1, self.num_head // self.num_kv_head)
        if not force and bs * self.num_kv_head * ((group + 15) // 16) <= 2 * self.device_core_count:
            return None   # (force: a split schedule's work items are dealt longest request first whatever the batch size)
        if not use_graph_bufs and fb.seq_lens_cpu is not None:
            cpu = fb.seq_lens_cpu
            if int(cpu.max()) == int(cpu.min()):
                return None
        order = torch.argsort(fb.seq_lens[:bs], descending=True).to(torch.int32)
        if use_graph_bufs:
            buf = self._graph["request_order"]  # address-stable: allocated by init_cuda_graph_state
            buf[:bs].copy_(order)
            return buf[:bs]
        return order

    def _plan_cascade_groups(self, fb: ForwardBatch, bs: int):
        """Shared-prefix groups of this decode batch, when the scheduler handed the radix nodes (or the groups) over:
        None unless they describe something the batch-wide device plan (ops.CascadeDecode) cannot -- more than one group,
        or a group that is only part of the batch."""
        groups = getattr(fb, "cascade_groups", None)
        nodes = getattr(fb, "radix_last_nodes", None)
        lens = None if fb.seq_lens_cpu is None else fb.seq_lens_cpu.tolist()
        if groups is None and nodes is not None and self._kv_head_dim_ok_for_groups():
            from ..mem_cache.radix_cache import plan_shared_prefix_groups
            # the tree plan is O(bs * depth) host Python: re-used while the batch's nodes (and the lengths' floor,
            # which only grows during decode) are the ones it was made for
            key = tuple(getattr(n, "id", None) if getattr(n, "id", None) is not None else id(n) for n in nodes)
            cached = self._groups_plan_cache
            if cached is not None and cached[0] == key and (lens is None or all(
                    L < min(lens[i] for i in m) for m, L in cached[1])):
                groups = cached[1]
            else:
                groups = plan_shared_prefix_groups(nodes, lens, min_shared=self.cascade_min_shared,
                                                   min_members=self.cascade_min_members)
                self._groups_plan_cache = (key, groups)
        elif groups is not None and lens is not None:
            # caller-supplied groups: a shared length must leave every member a non-empty suffix (the step's own row)
            checked = []
            for m, L in groups:
                m = [int(i) for i in m]
                if any(i < 0 or i >= bs for i in m):
                    raise ValueError(f"cascade_groups: member rows {m} outside the batch of {bs}")
                L = min(int(L), min(lens[i] for i in m) - 1)
                if L >= 1 and len(m) >= 1:
                    checked.append((m, L))
            groups = checked
        if not groups or not self._kv_head_dim_ok_for_groups():
            return None
        groups = groups[: 32]
        if len(groups) == 1 and len(groups[0][0]) == bs:
            return None
        return groups

    def _kv_head_dim_ok_for_groups(self) -> bool:
        kb = self.token_to_kv_pool.get_key_buffer(getattr(self.token_to_kv_pool, "start_layer", 0))
        return (kb.shape[-1] in (64, 96, 128, 256) and kb.shape[-1] == self.v_head_dim
                and kb.dtype in (torch.bfloat16, torch.float16))

    def _fill_num_kv_splits(self, out: torch.Tensor, lens: torch.Tensor, max_kv_splits: Optional[int] = None):
        """get_num_kv_splits (triton_backend.py:303-349): the reference's K3 formula, or -- deterministic inference --
        ceil(len / split_tile_size) per request (rows of a top-k group share their request's count)."""
        if self.enable_deterministic:
            n = ((lens + (self.split_tile_size - 1)) // self.split_tile_size).to(torch.int32)
            group = out.shape[0] // max(1, lens.shape[0])
            out.copy_(n.repeat_interleave(group) if group > 1 else n)
            return
        cap = self.max_kv_splits if max_kv_splits is None else max_kv_splits
        if self.static_kv_splits or self.device_core_count <= 0:  # SGLANG_TRITON_DECODE_ATTN_STATIC_KV_SPLITS (:215-217, :321-325)
            out.fill_(cap)
            return
        ops.get_num_kv_splits(out, lens, self.num_head, self.num_kv_head, cap, self.device_core_count)

    def _decode_metadata_draft(self, fb: ForwardBatch, use_graph_bufs: bool) -> ForwardMetadata:
        """Multi-step draft decode (triton_backend.py:772-774, :587-602): the page tables come from spec_info -- one row per
        (request, top-k branch), built by HipRadixMultiStepDraftBackend -- so the batch the kernel sees has
        kv_indptr.shape[0] - 1 rows; the split schedule is the reference's K3 formula with num_group = topk."""
        spec = fb.spec_info
        kv_indptr, kv_indices = spec.kv_indptr, spec.kv_indices
        rows = kv_indptr.shape[0] - 1
        S = self.max_kv_splits
        dg = getattr(self, "_draft_graph", None)
        if use_graph_bufs and dg is not None:
            if rows > dg["max_rows"]:
                raise ValueError(f"draft decode: {rows} rows exceed init_cuda_graph_state's max_num_tokens {dg['max_rows']}")
            num_kv_splits = dg["num_kv_splits"][:rows]
            attn_logits, attn_lse = dg["attn_logits"][:rows], dg["attn_lse"][:rows]
        else:
            num_kv_splits = torch.empty((rows,), dtype=torch.int32, device=self.device)
            attn_logits, attn_lse = self._scratch(rows, S)
        num_seqs = fb.seq_lens.shape[0]
        if rows % max(num_seqs, 1) == 0 and num_seqs > 0:
            self._fill_num_kv_splits(num_kv_splits, fb.seq_lens, S)
        else:  # (rows that are not whole top-k groups: every row keeps the full cap; unused slots cost a dead workgroup)
            num_kv_splits.fill_(S)
        return ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices, None, S, draft=True)

    def _decode_metadata(self, fb: ForwardBatch, bs: int, use_graph_bufs: bool) -> ForwardMetadata:
        if fb.spec_info is not None and getattr(fb.spec_info, "kv_indptr", None) is not None:
            return self._decode_metadata_draft(fb, use_graph_bufs)
        # (latent MLA rows are shared by all heads already: the cascade pays from ~128 requests on -- 64 x (3584 shared +
        # 512 own): 54 -> 78 us per layer, 256: 194 -> 122, 256 x (8192 + 256): 348 -> 158; tools/cascade_bench.py MLA=1)
        self._cascade_on = self.cascade_decode and bs >= (max(self.cascade_min_bs, 128) if self._is_mla_pool
                                                          else self.cascade_min_bs)
        self._cascade_call = None
        groups = (self._plan_cascade_groups(fb, bs)
                  if (self.cascade_decode and not self._is_mla_pool and not use_graph_bufs) else None)  # (geometry varies: eager steps only)
        if groups:
            # several shared prefixes (one per radix-tree node), or one that only part of the batch shares
            if self._cascade_groups is None:
                kb = self.token_to_kv_pool.get_key_buffer(getattr(self.token_to_kv_pool, "start_layer", 0))
                try:
                    self._cascade_groups = ops.CascadeGroups(
                        self.req_to_token_pool.size, self.num_head, self.num_kv_head, kb.shape[-1], self._q_dtype(kb),
                        self.device, max_shared_total=max(self.req_to_token.shape[1], self.cascade_groups_max_shared),
                        cu_count=self.device_core_count, max_kv_splits=self.native_split_cap)
                except torch.OutOfMemoryError:
                    # created on the first grouped step, after the KV pool took its share of the memory: without room
                    # for the gathered queries / partials the batch takes the plain path (same results)
                    self._cascade_groups = False
            if self._cascade_groups:
                self._cascade_groups.plan(self.req_to_token, fb.req_pool_indices, fb.seq_lens, groups)
                self._cascade_on, self._cascade_call = True, self._cascade_groups
                return ForwardMetadata(None, None, None, None, None, None, None, 1)
        if self._cascade_on:
            if self._cascade is None:
                kb = self.token_to_kv_pool.get_key_buffer(getattr(self.token_to_kv_pool, "start_layer", 0))
                mla = (self._is_mla_pool and kb.shape[-1] == 576 and self.v_head_dim == 512
                       and kb.dtype in (torch.bfloat16, torch.float16))   # 16-bit latent rows (phase 1 reads them)
                d256 = kb.shape[-1] in (96, 256) and kb.dtype in (torch.bfloat16, torch.float16)   # (no fp8 kernels at 96 / 256)
                if not mla and (self._is_mla_pool or not (kb.shape[-1] in (64, 128) or d256) or kb.shape[-1] != self.v_head_dim):
                    self.cascade_decode = self._cascade_on = False
                else:
                    self._cascade = ops.CascadeDecode(
                        self.req_to_token_pool.size, self.num_head, self.num_kv_head, kb.shape[-1],
                        self._q_dtype(kb), self.device,
                        max_shared=self.req_to_token.shape[1], cu_count=self.device_core_count,
                        min_shared=self.cascade_min_shared, max_kv_splits=self.native_split_cap,
                        v_head_dim=self.v_head_dim)
            if self._cascade_on:
                self._cascade.plan(self.req_to_token, fb.req_pool_indices, fb.seq_lens)
                self._cascade_call = self._cascade
                return ForwardMetadata(None, None, None, None, None, None, None, 1)
        if self.split_policy == "native" and self.sliding_window_size is None:
            return self._decode_metadata_native(fb, bs, use_graph_bufs)
        splits_needed = True
        if self.enable_deterministic or self.static_kv_splits:
            splits_needed = self.max_kv_splits > 1  # (one launch shape for every batch)
        elif not use_graph_bufs and fb.seq_lens_cpu is not None and self.max_kv_splits > 1:
            host = host_num_kv_splits(np.asarray(fb.seq_lens_cpu), self.num_head, self.num_kv_head,
                                      self.max_kv_splits, self.device_core_count)
            splits_needed = bool(host.max() > 1)
        elif self.max_kv_splits <= 1:
            splits_needed = False
        kv_indptr = kv_indices = None
        if self.decode_index_mode == "indices":
            if use_graph_bufs:
                kv_indices = self._graph_kv_indices()
            else:
                total = fb.seq_lens_sum if fb.seq_lens_sum is not None else bs * self.max_context_len
                kv_indices = torch.empty(total, dtype=torch.int64, device=self.device)
            kv_indptr = ops.build_kv_indices(self.req_to_token, fb.req_pool_indices, fb.seq_lens,
                                             self.kv_indptr, kv_indices)
        win = {}
        if self.sliding_window_size is not None:
            wp, wi, wl, _ = self._window(fb.seq_lens, fb.req_pool_indices, bs, use_graph_bufs)
            wsplits = (self._graph["window_num_kv_splits"][:bs] if use_graph_bufs
                       else torch.empty((bs,), dtype=torch.int32, device=self.device))
            self._fill_num_kv_splits(wsplits, wl)
            win = dict(window_kv_indptr=wp, window_kv_indices=wi, window_num_kv_splits=wsplits)
            splits_needed = splits_needed or self.max_kv_splits > 1  # window launch uses the scratch too
        if not splits_needed:
            return ForwardMetadata(None, None, None, None, kv_indptr, kv_indices, None, 1, **win,
                                   request_order=None if win else self._request_order(fb, bs, use_graph_bufs))
        if use_graph_bufs:
            num_kv_splits = self._graph["num_kv_splits"][:bs]
            attn_logits, attn_lse = self._graph["attn_logits"][:bs], self._graph["attn_lse"][:bs]
        else:
            num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
            attn_logits, attn_lse = self._scratch(bs)
        self._fill_num_kv_splits(num_kv_splits, fb.seq_lens)
        return ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices,
                               None, self.max_kv_splits, **win)

    def _decode_metadata_native(self, fb: ForwardBatch, bs: int, use_graph_bufs: bool) -> ForwardMetadata:
        """Split schedule sized for the chip, not by the reference's formula: S depends on bs only, so it is
        graph-stable; the per-request counts (short requests get fewer splits) are written on the device."""
        kv_indptr = kv_indices = None
        if self.decode_index_mode == "indices":
            if use_graph_bufs:
                kv_indices = self._graph_kv_indices()
            else:
                total = fb.seq_lens_sum if fb.seq_lens_sum is not None else bs * self.max_context_len
                kv_indices = torch.empty(total, dtype=torch.int64, device=self.device)
            kv_indptr = ops.build_kv_indices(self.req_to_token, fb.req_pool_indices, fb.seq_lens,
                                             self.kv_indptr, kv_indices)
        group = max(1, self.num_head // self.num_kv_head)
        blocks = bs * self.num_kv_head * ((group + 15) // 16)
        wg_target = self.device_core_count * 2
        # Dense kernel: the LENGTH-AWARE schedule (rx_num_kv_splits_balanced).  A batch with fewer whole-request
        # workgroups than ~0.7 per CU is cut to two workgroups per CU in total (16 x 4 k 72 -> 53 us per layer; 8 x 8 k
        # 127 -> 54; beyond that every split only costs); a near-uniform batch of 0.7 - 3 per CU goes by the FILL rule
        # (round 4, radix_hip.h: whole up to one per CU -- TP=8 shard 256 x 4 k 105 -> 100 us, 192 x 4 k 93 -> 76 -- and
        # above it the count whose pieces fill whole rounds of CUs: 320 x 4 k 153 -> 125).  Otherwise a request takes
        # ceil(len / t*) splits, t* = the batch's even share per workgroup -- but ONLY if it is well above that share:
        # in a uniform batch nobody is split (bs 256 x 4 k: one pass each), while one 32 k-token request among 63 of 1 k
        # is cut ~20 ways instead of being the kernel's tail (516 -> 139 us per layer).  Requests with one split write
        # their output straight from stage 1 (rx_decode_params: direct_single).  In the 2-4 split regime a split under
        # ~1 k tokens is all prologue (32 x 1 k: 31 us at 1 split, 33 at 2): t* has a 1 k floor there, 128 for tiny batches.
        min_tokens = 1024 if (2 * blocks >= self.device_core_count and not self._is_mla_pool) else 128
        cap = self.native_split_cap
        # a MIXED batch (some requests cut, some whole) takes the ROUNDS rule of rx_num_kv_splits_balanced
        # (wg_target_mixed = -1): long requests are cut into pieces that last as many rounds of workgroups as the unsplit
        # ones need -- made for the kernel's usual two workgroups per CU, so eager and graph-replayed steps share it
        # (one 32 k request among 63 of 1 k: 94 -> 80 us per layer at two per CU).  RX_SPLIT_OCC3=1 (eager steps only):
        # the three-per-CU kernel instance with its own schedule (3 x CUs near-equal pieces) instead -- the same times
        # within 2 % on the batches of tools/probe/rounds_rule.py.
        use_items = not self._is_mla_pool and not self._no_split_items
        three = use_items and self._split_occ3 and not use_graph_bufs and fb.seq_lens_cpu is not None
        # (an MLA pool: no live-pairs grid and no mixed-batch rule -- wg_target_mixed = wg_target switches those off and
        # selects the fill rule's whole-requests form: from 0.8 requests per CU up nobody is cut.  256 requests x 4 k latent
        # rows one pass each 122 us, cut in two 135 (fp8 rows; 16-bit 216 vs 235); 224 x 4 k 113 vs 127 cut in three;
        # tools/probe/mla_split_sweep.py)
        wg_mixed = self.device_core_count * 3 if three else (-1 if use_items else wg_target)
        host_pairs = None
        if not use_graph_bufs and fb.seq_lens_cpu is not None:
            host_counts = ops.balanced_kv_splits_host(fb.seq_lens_cpu.numpy()[:bs], self.num_head, self.num_kv_head, cap,
                                                      wg_target, min_tokens, wg_mixed)
            S = int(host_counts.max())
            host_pairs = int(host_counts.sum())
        else:  # lengths unknown here (graph replay refills the counts on the device): slots by batch size alone
            S = self._graph_split_slots(bs)
        if S <= 1:
            order1 = self._request_order(fb, bs, use_graph_bufs)
            return ForwardMetadata(None, None, None, None, kv_indptr, kv_indices, None, 1, request_order=order1,
                                   decode_units=self._build_decode_units(fb, bs, use_graph_bufs, None, 1, None, order1, kv_indices))
        S_cap = S
        # the in-kernel stage 2 (merge_counters) wants the partial rows of a head in chunks of 8: round the slots up while
        # the partials fit its bound -- the surplus workgroups exit at once, and the second launch goes
        wgpr = self.num_kv_head * ((group + 15) // 16)
        # split requests share ~wg_target workgroups: (request, split) pairs with a partial <= that / blocks per request,
        # with slack for rounding up -- what the in-kernel stage 2's size bound looks at, not bs * slots
        # (the fill rule hands a near-uniform batch ONE count for everybody: the exact host count when the lengths are
        # known, else the same bound the graph path sizes its table by -- ADVICE r4)
        pairs = (int(host_counts[host_counts > 1].sum()) if host_pairs is not None  # (a request with one split writes none)
                 else self._split_pairs_bound(bs, S_cap))
        if self._merge_counters is not None and S % 8:
            S8 = (S + 7) // 8 * 8
            if min(bs * S8, pairs) * self.num_head * self.v_head_dim * 4 <= (4 << 20):
                S = S8
        pairs = min(bs * S, pairs)
        if use_graph_bufs:
            num_kv_splits = self._graph["num_kv_splits"][:bs]
            n = bs * self.num_head * S
            attn_logits = self._graph["native_logits"][: n * self.v_head_dim].view(bs, self.num_head, S, self.v_head_dim)
            attn_lse = self._graph["native_lse"][:n].view(bs, self.num_head, S)
        else:
            num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
            attn_logits, attn_lse = self._scratch(bs, S)
        # The device schedule takes the cap the HOST mirror ran with: under the rounds rule the cap feeds
        # the round tests, so a device pass with the smaller cap S_cap = max(host counts) can stop at a smaller R and hand
        # out MORE pairs than the host counted -- the table and grid sized from host_pairs would then leave the shortest
        # requests without a workgroup.  With one cap the two are the same integer arithmetic (max = S_cap <= S slots).
        ops.get_num_kv_splits_balanced(num_kv_splits, fb.seq_lens, self.num_head, self.num_kv_head,
                                       cap if host_pairs is not None else S_cap, wg_target,
                                       min_tokens_per_split=min_tokens, wg_target_mixed=wg_mixed)
        order = self._request_order(fb, bs, use_graph_bufs, force=True)
        # the grid holds the LIVE (request, split) pairs only, longest requests first (rx_decode_params.split_items): with
        # split slots the long request's later splits sit behind hundreds of dead workgroups in dispatch order
        items = None
        if use_items:
            if use_graph_bufs:
                # the cap is a BOUND here (the counts are refilled on the device before every replay): the guarded build
                # replaces a schedule that outgrows it by whole-request passes instead of dropping pairs
                items = self._graph["split_items"].build(num_kv_splits, order, cap=self._split_pairs_bound(bs, S_cap),
                                                         guarded=True)
            else:
                mixed = three and bool((host_counts > 1).any() and (host_counts == 1).any())
                items = ops.SplitItems(host_pairs if host_pairs is not None else bs * S_cap, self.device).build(
                    num_kv_splits, order, wgs_per_cu=3 if mixed else 0)
        if items is not None and self._debug_checks and not torch.cuda.is_current_stream_capturing():  # RX_DEBUG_CHECKS=1 (a host sync): the table holds every live pair
            live = int(items.count.item())
            if live > items.cap:
                raise AssertionError(f"split items: {live} live (request, split) pairs > the table's {items.cap}")
        return ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices, None, S,
                               request_order=order, partial_pairs_hint=pairs, split_items=items,
                               decode_units=self._build_decode_units(fb, bs, use_graph_bufs, num_kv_splits, S, items, order, kv_indices))

    def _build_decode_units(self, fb, bs, use_graph_bufs, num_kv_splits, S, items, order, kv_indices):
        """The per-unit tables of this forward's decode launches (ops.DecodeUnits, rx_decode_units): req_to_token mode of
        the dense MFMA kernel only -- whole requests, or the live-pairs grid of a split schedule.  Under graph replay the
        tables are address-stable buffers refilled here before every replay, like every other piece of metadata."""
        if (self._no_decode_units or kv_indices is not None or self.decode_index_mode != "paged" or self._is_mla_pool
                or self.dcp is not None or not self._decode_honours_split_items() or (S > 1 and items is None)):
            return None
        if use_graph_bufs:
            units = self._graph["decode_units"]
        else:
            units = ops.DecodeUnits(bs if items is None else items.cap, self.device)
        return units.build(self.req_to_token, fb.req_pool_indices, fb.seq_lens, num_kv_splits, S, items, order)

    def _split_pairs_bound(self, bs: int, slots: int) -> int:
        """An upper bound of the (request, split) pairs rx_num_kv_splits_balanced can hand out to ``bs`` requests with
        ``slots`` as its cap, whatever the lengths (split_pairs_bound below; tests/test_cabi_and_host.py sweeps the host
        mirror against it)."""
        group = max(1, self.num_head // self.num_kv_head)
        return split_pairs_bound(bs, slots, self.num_kv_head * ((group + 15) // 16), self.device_core_count)

    def _decode_honours_split_items(self) -> bool:
        """rx_decode_attn reads the (request, split) table only in its MFMA kernel (head dims 64 / 96 / 128 / 256 with
        equal k and v dims); every other layer shape replays the bs x slots grid, where an unused slot IS a dead
        workgroup."""
        try:
            kb = self.token_to_kv_pool.get_key_buffer(getattr(self.token_to_kv_pool, "start_layer", 0))
            dk = int(kb.shape[-1])
        except Exception:
            return False
        return dk == self.v_head_dim and dk in (64, 96, 128, 256)

    def _graph_split_slots(self, bs: int) -> int:
        """Split slots of a graph-replayed dense decode step: enough for a small batch to fill the chip, and at least 8
        so that a long request in a large batch can still be cut (the counts themselves are refilled on the device
        before every replay; a request with one split costs nothing, see direct_single)."""
        group = max(1, self.num_head // self.num_kv_head)
        blocks = max(1, bs * self.num_kv_head * ((group + 15) // 16))
        S = max(8, min(self.native_split_cap, -(-2 * self.device_core_count // blocks)))
        if not self._is_mla_pool and not self._no_split_items and self._decode_honours_split_items():
            # the live-pairs grid: a slot nobody uses costs nothing (no dead workgroup), so a batch keeps the full cap --
            # the long request of a mixed batch is cut as finely under graph replay as in an eager step -- while the
            # fp32 partials [bs, Hq, slots, Dv] stay inside graph_partials_budget (512 requests x 64 heads x
            # 32 slots x 128 x 4 B = 537 MB): halve the slots (never below 8) until they fit
            S_full = max(S, self.native_split_cap)
            while S_full > max(S, 8) and bs * self.num_head * S_full * self.v_head_dim * 4 > self.graph_partials_budget:
                S_full //= 2
            S = max(S, S_full)
        if self._merge_counters is not None and S % 8 and bs * self.num_head * ((S + 7) // 8 * 8) * self.v_head_dim * 4 <= (4 << 20):
            S = (S + 7) // 8 * 8
        return S

    # ------------------------------------------------------------------ decode context parallel
    def _decode_metadata_dcp(self, fb: ForwardBatch, bs: int, use_graph_bufs: bool) -> ForwardMetadata:
        """_update_decode_kv_buffers' DCP branch (triton_backend.py:422-433): the rank's local kv_indices, and the
        split schedule from the per-rank lengths (clamped to >= 1).  Always two stages: the cross-rank join works on
        the fp32 partials."""
        d = self.dcp
        if use_graph_bufs:
            kv_indices = self._graph_kv_indices()
        else:
            total = fb.seq_lens_sum if fb.seq_lens_sum is not None else bs * self.max_context_len
            kv_indices = torch.empty(total // d.size + bs + 1, dtype=torch.int64, device=self.device)
        dcp_lens = torch.empty((bs,), dtype=torch.int32, device=self.device)
        kv_indptr = ops.dcp_kv_indices(self.req_to_token, fb.req_pool_indices, fb.seq_lens, self.kv_indptr, kv_indices,
                                       d.size, d.rank, dcp_lens=dcp_lens)
        S = max(2, self.max_kv_splits)
        if use_graph_bufs:
            num_kv_splits = self._graph["num_kv_splits"][:bs]
            n = bs * self.num_head * S
            attn_logits = self._graph["native_logits"][: n * self.v_head_dim].view(bs, self.num_head, S, self.v_head_dim)
            attn_lse = self._graph["native_lse"][:n].view(bs, self.num_head, S)
        else:
            num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
            attn_logits, attn_lse = self._scratch(bs, S)
        ops.get_num_kv_splits(num_kv_splits, dcp_lens.clamp_(min=1), self.num_head, self.num_kv_head, S,
                              self.device_core_count)
        return ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices, None, S)

    def _extend_metadata_dcp(self, fb: ForwardBatch, bs: int) -> ForwardMetadata:
        """The cached prefix through the rank's local kv_indices (triton_backend.py:728-737); qo_indptr as ever."""
        d = self.dcp
        if fb.extend_prefix_lens_cpu is not None:
            total = int(sum(fb.extend_prefix_lens_cpu))
        else:
            total = bs * self.max_context_len
        kv_indices = torch.empty(total // d.size + bs + 1, dtype=torch.int64, device=self.device)
        kv_indptr = ops.dcp_kv_indices(self.req_to_token, fb.req_pool_indices, fb.extend_prefix_lens, self.kv_indptr,
                                       kv_indices, d.size, d.rank)
        qo_indptr = self.qo_indptr[: bs + 1]
        qo_indptr[1:] = torch.cumsum(fb.extend_seq_lens, dim=0)
        if fb.extend_seq_lens_cpu is not None:
            max_extend_len = max(fb.extend_seq_lens_cpu)
        else:
            max_extend_len = int(fb.extend_seq_lens.max())
        self._extend_split_on = False
        md = ForwardMetadata(None, None, max_extend_len, None, kv_indptr, kv_indices, qo_indptr)
        md.dcp_prefix_total = total  # the WHOLE prefix, known to every rank alike: decides the collective path
        return md

    def _loc_info(self, loc: torch.Tensor):
        """What set_kv_buffer gets as its write location (triton_backend.py:1287-1293): the pool's own KVWriteLoc
        around the out_cache_loc tensor, or the bare tensor when the pool's module has no such class."""
        return self._write_loc_cls(loc) if self._write_loc_cls is not None else loc

    def _dcp_store(self, layer, fb: ForwardBatch, k, v):
        """_set_kv_buffer's DCP branch (triton_backend.py:1227-1239): the rank stores the tokens it owns, at the
        local slot; the other rows carry the pool's skip index."""
        if fb.positions is None or fb.positions.numel() != fb.out_cache_loc.numel():
            raise ValueError("DCP needs forward_batch.positions for the new tokens")
        loc = ops.dcp_store_loc(fb.out_cache_loc, fb.positions, self.dcp.size, self.dcp.rank, skip_index=0)
        self.token_to_kv_pool.set_kv_buffer(layer, self._loc_info(loc), k, v, layer.k_scale, layer.v_scale)

    def _forward_decode_dcp(self, q3, o3, layer, fb: ForwardBatch, sinks):
        """triton_backend.py:1797-1839: gathered q heads over the local tokens (kv-split partials only), the rank's
        fp32 result and LSE, then cp_lse_ag_out_rs_mha."""
        if sinks is not None or (getattr(layer, "xai_temperature_len", -1) or 0) > 0:
            raise NotImplementedError("DCP decode: sinks and the Grok temperature need the whole sequence on one rank")
        md = self.forward_metadata
        bs = q3.shape[0]
        q_all = self.dcp.all_gather_heads(q3.contiguous())
        k_descale, v_descale = self._scales(layer)
        k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        hnd = getattr(self.token_to_kv_pool, "use_hnd", False)
        attn_logits, attn_lse = md.attn_logits[:bs], md.attn_lse[:bs]
        attn_lse.fill_(float("-inf"))  # splits that do not run (and ranks without tokens) stay empty
        o_unused = q_all.new_empty(bs, q_all.shape[1], layer.v_head_dim)
        ops.decode_attention_fwd(q_all, k_buf, v_buf, o_unused, md.kv_indptr, md.kv_indices, attn_logits, attn_lse,
                                 md.num_kv_splits, md.max_kv_splits, layer.scaling, k_descale, v_descale,
                                 logit_cap=layer.logit_cap, page_size=self.page_size,
                                 kv_layout=ops.kv_layout_hnd(k_buf, v_buf) if hnd else None, stages=1)
        o32, lse = ops.dcp_local_merge(attn_logits, attn_lse, v_scale=v_descale)
        self.dcp.merge_partials(o32, lse, o3)

    def _forward_extend_dcp(self, q3, k3, v3, o3, layer, fb: ForwardBatch, causal: bool, sinks):
        """_forward_extend_dcp (triton_backend.py:1439-1569): the new tokens' own block with the local heads (their
        K/V are whole on every rank), the cached prefix with the gathered heads over the local tokens, joined across
        the ranks by LSE and then with the first part.  Partials are 16-bit (what the extend kernels write, as every
        merge_state input in the reference is)."""
        if sinks is not None or md_has_mask(self.forward_metadata):
            raise NotImplementedError("DCP extend: no sinks / custom masks (as the reference)")
        if layer.sliding_window_size is not None and layer.sliding_window_size > -1:
            raise NotImplementedError("DCP extend: no sliding window (as the reference)")
        md = self.forward_metadata
        T, h_loc = q3.shape[0], q3.shape[1]
        k_descale, v_descale = self._scales(layer)
        k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        lay = ops.kv_layout_hnd(k_buf, v_buf) if getattr(self.token_to_kv_pool, "use_hnd", False) else None
        xai = layer.xai_temperature_len
        empty_indptr = torch.zeros_like(md.kv_indptr)
        has_prefix = md.dcp_prefix_total > 0
        cur_o = torch.empty_like(o3) if has_prefix else o3
        cur_lse = torch.full((T, h_loc), float("-inf"), dtype=torch.float32, device=q3.device)
        ops.extend_attention_fwd(q3, k3, v3, cur_o, k_buf, v_buf, md.qo_indptr, empty_indptr, md.kv_indices[:0], None,
                                 causal, None, md.max_extend_len, 1.0, 1.0, sm_scale=layer.scaling,
                                 logit_cap=layer.logit_cap, xai_temperature_len=xai, lse_extend=cur_lse,
                                 skip_prefix=True, page_size=self.page_size, kv_layout=lay)
        if not has_prefix:
            return
        q_all = self.dcp.all_gather_heads(q3.contiguous())
        h_all = q_all.shape[1]
        pre_o = torch.empty(T, h_all, layer.v_head_dim, dtype=q3.dtype, device=q3.device)
        pre_lse = torch.full((T, h_all), float("-inf"), dtype=torch.float32, device=q3.device)
        # skip_extend: the new tokens' K/V are not read (one-row stand-ins carry the head count and dims)
        k_none = q3.new_empty(1, layer.tp_k_head_num, layer.qk_head_dim)
        v_none = q3.new_empty(1, layer.tp_v_head_num, layer.v_head_dim)
        ops.extend_attention_fwd(q_all, k_none, v_none, pre_o, k_buf, v_buf, md.qo_indptr, md.kv_indptr, md.kv_indices,
                                 None, False, None, md.max_extend_len, k_descale, v_descale, sm_scale=layer.scaling,
                                 logit_cap=layer.logit_cap, xai_temperature_len=xai, lse_extend=pre_lse,
                                 skip_extend=True, page_size=self.page_size, kv_layout=lay)
        self.dcp.merge_partials(ops.dcp_widen(pre_o), pre_lse, o3, cur_o, cur_lse)

    def _extend_metadata(self, fb: ForwardBatch, bs: int) -> ForwardMetadata:
        # prefix-only kv indices + qo_indptr (triton_backend.py:869-924)
        if fb.extend_prefix_lens_cpu is not None:
            total = int(sum(fb.extend_prefix_lens_cpu))
        else:
            total = bs * self.max_context_len
        kv_indices = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
        kv_indptr = ops.build_kv_indices(self.req_to_token, fb.req_pool_indices,
                                         fb.extend_prefix_lens, self.kv_indptr, kv_indices)
        qo_indptr = self.qo_indptr[: bs + 1]
        qo_indptr[1:] = torch.cumsum(fb.extend_seq_lens, dim=0)
        if fb.extend_seq_lens_cpu is not None:
            max_extend_len = max(fb.extend_seq_lens_cpu)
        else:
            max_extend_len = int(fb.extend_seq_lens.max())
        win = {}
        if self.sliding_window_size is not None:  # window over the cached prefix (triton_backend.py:891-905)
            wp, wi, _, wo = self._window(fb.extend_prefix_lens, fb.req_pool_indices, bs)
            win = dict(window_kv_indptr=wp, window_kv_indices=wi, window_kv_offsets=wo)
        # a small batch of SHORT extends over LONG cached prefixes (a follow-up turn on a long conversation): one
        # workgroup per (request, kv head, query block) leaves most CUs idle and each walks the whole prefix -- the
        # split-KV form of the verify path, with the causal rule in place of a tree mask
        self._extend_split_on = False
        ext, pre = fb.extend_seq_lens_cpu, fb.extend_prefix_lens_cpu
        if (self.sliding_window_size is None and self._split_dims() is not None and ext is not None
                and pre is not None and len(set(int(e) for e in ext)) == 1 and 0 < int(ext[0]) <= 512
                and min(int(p) for p in pre) >= 1024):
            if self._verify_split is None:
                kb = self.token_to_kv_pool.get_key_buffer(getattr(self.token_to_kv_pool, "start_layer", 0))
                dk, dv = self._split_dims()
                self._verify_split = ops.VerifySplitKV(self.num_head, self.num_kv_head, self._q_dtype(kb), self.device,
                                                       cu_count=self.device_core_count, head_dim=dk, v_head_dim=dv)
            if self._verify_split.num_chunks(bs, int(ext[0])) >= 2:
                self._verify_split.plan(qo_indptr, kv_indptr, kv_indices, None, None, int(ext[0]))
                self._extend_split_on = True
        return ForwardMetadata(None, None, max_extend_len, None, kv_indptr, kv_indices, qo_indptr, **win)

    def _target_verify_metadata(self, fb: ForwardBatch, bs: int, graph: bool = False) -> ForwardMetadata:
        """TARGET_VERIFY (triton_backend.py:801-866): every request extends by its draft tokens over its
        WHOLE cached sequence, under the draft tree's mask.  ``graph``: qo_indptr / kv_indices / mask_indptr
        live in the address-stable buffers (triton_backend.py:1016-1063), refilled before every replay."""
        spec = fb.spec_info
        nd = self.num_draft_tokens
        if spec is not None and getattr(spec, "draft_token_num", None) is not None:
            nd = int(spec.draft_token_num)
        if not nd:
            raise ValueError("TARGET_VERIFY needs spec_info.draft_token_num or server_args.speculative_num_draft_tokens")
        custom_mask = getattr(spec, "custom_mask", None)
        if custom_mask is None:
            raise ValueError("TARGET_VERIFY needs spec_info.custom_mask (the draft tree's visibility mask)")
        qo_indptr = self.qo_indptr[: bs + 1]
        torch.arange(0, (1 + bs) * nd, step=nd, dtype=torch.int64, device=self.device, out=qo_indptr)
        if graph:
            kv_indices = self._graph_kv_indices()
        else:
            total = fb.seq_lens_sum if fb.seq_lens_sum is not None else bs * self.max_context_len
            kv_indices = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
        kv_indptr = ops.build_kv_indices(self.req_to_token, fb.req_pool_indices, fb.seq_lens, self.kv_indptr,
                                         kv_indices)
        win = {}
        if self.sliding_window_size is not None:
            wp, wi, _, wo = self._window(fb.seq_lens, fb.req_pool_indices, bs, graph)
            win = dict(window_kv_indptr=wp, window_kv_indices=wi, window_kv_offsets=wo)
        mask_indptr = self.mask_indptr[: bs + 1]
        mask_indptr[1:] = torch.cumsum(nd * (fb.seq_lens[:bs].to(torch.int64) + nd), dim=0)
        if graph:
            # the captured kernels read the mask bytes in place: keep them at one address (a runner's bool mask
            # would be re-cast into a fresh uint8 tensor on every forward)
            g = self._graph
            need = g["max_bs"] * nd * (self.max_context_len + nd)
            if g.get("custom_mask") is None:
                if g.get("captured"):
                    raise RuntimeError("TARGET_VERIFY under graph replay needs server_args.speculative_num_draft_tokens "
                                       "at init_cuda_graph_state time (the mask buffer must exist before capture)")
                g["custom_mask"] = torch.zeros(need, dtype=torch.uint8, device=self.device)
            if custom_mask.numel() > g["custom_mask"].numel() or need > g["custom_mask"].numel():
                # never re-allocated: captured kernels hold the address
                raise ValueError("TARGET_VERIFY mask larger than the graph state's max_bs * nd * (context_len + nd)")
            g["custom_mask"][: custom_mask.numel()].copy_(custom_mask.reshape(-1))
            custom_mask = g["custom_mask"]
        # small verify batches: one workgroup per (request, kv head) would leave the chip idle -- cut the cached
        # part into chunks (ops.VerifySplitKV, the reference's verify_splitkv case)
        self._verify_split_on = False
        if self.sliding_window_size is None and self._split_dims() is not None:
            if self._verify_split is None:
                kb = self.token_to_kv_pool.get_key_buffer(getattr(self.token_to_kv_pool, "start_layer", 0))
                dk, dv = self._split_dims()
                self._verify_split = ops.VerifySplitKV(self.num_head, self.num_kv_head, self._q_dtype(kb), self.device,
                                                       cu_count=self.device_core_count, head_dim=dk, v_head_dim=dv)
            if self._verify_split.num_chunks(bs, nd) >= 2:
                cm = custom_mask if custom_mask.dtype == torch.uint8 else custom_mask.to(torch.uint8)
                self._verify_split.plan(qo_indptr, kv_indptr, kv_indices, cm, mask_indptr, nd)
                self._verify_split_on = True
        return ForwardMetadata(None, None, nd, None, kv_indptr, kv_indices, qo_indptr,
                               custom_mask=custom_mask, mask_indptr=mask_indptr, **win)

    def _draft_extend_metadata(self, fb: ForwardBatch, bs: int) -> ForwardMetadata:
        """DRAFT_EXTEND_V2 (triton_backend.py:907-924): spec_info produces the prefill arguments; a mask, when
        one comes back, is addressed through mask_indptr = cumsum(extend_len * (cached_len + extend_len))."""
        kv_indices, kv_indptr, qo_indptr, custom_mask = fb.spec_info.generate_attn_arg_prefill(
            fb.req_pool_indices, fb.seq_lens, None, self.req_to_token)
        ext = qo_indptr[1:] - qo_indptr[:-1]
        max_extend_len = int(getattr(fb.spec_info, "num_tokens_per_req", 0)) or int(ext.max())
        mask_indptr = None
        if custom_mask is not None:
            mask_indptr = self.mask_indptr[: bs + 1]
            cached = (kv_indptr[1:] - kv_indptr[:-1]).to(torch.int64)
            mask_indptr[1:] = torch.cumsum(ext.to(torch.int64) * (cached + ext.to(torch.int64)), dim=0)
        return ForwardMetadata(None, None, max_extend_len, None, kv_indptr.to(torch.int32), kv_indices,
                               qo_indptr, custom_mask=custom_mask, mask_indptr=mask_indptr)

    # ------------------------------------------------------------------ graph support
    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int, kv_indices_buf: Optional[torch.Tensor] = None,
                              cuda_graph_num_kv_splits_buf: Optional[torch.Tensor] = None):
        """Address-stable buffers (triton_backend.py:962-1063).  Everything a captured kernel reads that
        init_forward_metadata_out_graph rewrites before a replay lives here.  kv_indices_buf / cuda_graph_num_kv_splits_buf
        (:970-1001): the multi-step draft backend's shared buffers -- this backend then only serves draft decode steps
        under graphs and keeps the fp32 partials for max_num_tokens branch rows."""
        dev = self.device
        if kv_indices_buf is not None:
            rows = int(max_num_tokens)
            self._draft_graph = {
                "max_rows": rows, "kv_indices": kv_indices_buf,
                "num_kv_splits": (cuda_graph_num_kv_splits_buf if cuda_graph_num_kv_splits_buf is not None
                                  else torch.full((rows,), self.max_kv_splits, dtype=torch.int32, device=dev)),
                "attn_logits": torch.zeros((rows, self.num_head, self.max_kv_splits, self.v_head_dim), dtype=torch.float32, device=dev),
                "attn_lse": torch.zeros((rows, self.num_head, self.max_kv_splits), dtype=torch.float32, device=dev)}
        self._graph = {
            "max_bs": int(max_bs),
            "num_kv_splits": torch.full((max_bs,), 1, dtype=torch.int32, device=dev),
            "attn_logits": torch.zeros((max_bs, self.num_head, self.max_kv_splits, self.v_head_dim),
                                       dtype=torch.float32, device=dev),
            "attn_lse": torch.zeros((max_bs, self.num_head, self.max_kv_splits), dtype=torch.float32, device=dev),
            # int64 slot list of the "indices" contract and of TARGET_VERIFY: built on first use (the paged decode
            # mode never materialises it)
            "kv_indices": None,
        }
        # native schedule: bs * S(bs) <= cu_count / wg_per_request + bs rows of partials
        group = max(1, self.num_head // self.num_kv_head)
        rows = (2 * self.device_core_count // (self.num_kv_head * ((group + 15) // 16)) + max_bs + 1) * self.num_head
        rows = max(rows, (1 << 20) // self.v_head_dim + 8)  # split slots rounded up to 8 while the partials fit 4 MiB
        # the length-aware schedule's slots: _graph_split_slots(bs) per request
        rows = max(rows, max(b * self._graph_split_slots(b) for b in range(1, max_bs + 1)) * self.num_head)
        self._graph["native_logits"] = torch.zeros(rows * self.v_head_dim, dtype=torch.float32, device=dev)
        self._graph["native_lse"] = torch.zeros(rows, dtype=torch.float32, device=dev)
        # every buffer a captured kernel reads lives here from the start (a buffer created lazily and re-allocated
        # after capture would leave the graph reading the old address): the launch order of ragged batches ...
        self._graph["request_order"] = torch.zeros(max_bs, dtype=torch.int32, device=dev)
        # ... the (request, split) pairs of the length-aware schedule
        self._graph["split_items"] = ops.SplitItems(
            max(self._split_pairs_bound(b, self._graph_split_slots(b)) for b in range(1, max_bs + 1)), dev)
        # ... the per-unit descriptor tables of the decode launches (a unit = a pair of that table, or a whole request)
        self._graph["decode_units"] = ops.DecodeUnits(max(max_bs, self._graph["split_items"].items.numel() // 2), dev)
        # ... and the draft tree's mask bytes of TARGET_VERIFY, sized from speculative_num_draft_tokens
        nd = int(self.num_draft_tokens or 0)
        if nd > 0:
            self._graph["custom_mask"] = torch.zeros(max_bs * nd * (self.max_context_len + nd), dtype=torch.uint8,
                                                     device=dev)
        if self.decode_index_mode == "indices":
            self._graph_kv_indices()
        if self.sliding_window_size is not None:
            w = min(self.sliding_window_size, self.max_context_len)
            self._graph["window_kv_indices"] = torch.zeros(max(1, max_bs * w), dtype=torch.int64, device=dev)
            self._graph["window_kv_offsets"] = torch.zeros(max_bs, dtype=torch.int32, device=dev)
            self._graph["window_num_kv_splits"] = torch.ones(max_bs, dtype=torch.int32, device=dev)

    def _graph_kv_indices(self) -> torch.Tensor:
        g = self._graph
        if g["kv_indices"] is None:
            g["kv_indices"] = torch.zeros((g["max_bs"] * self.max_context_len,), dtype=torch.int64, device=self.device)
        return g["kv_indices"]

    def get_cuda_graph_seq_len_fill_value(self):
        return 1  # triton_backend.py:1205-1206

    def on_after_cuda_graph_warmup(self):
        pass

    def draft_extend_metadata_captured_in_graph(self) -> bool:
        return False  # base_attn_backend.py:94-99: replay metadata is rebuilt eagerly, out of graph

    def update_verify_buffers_to_fill_after_draft(self, spec_info, cuda_graph_bs):
        pass  # triton_backend.py:1212-1215: the tree mask is read in place, nothing derived from it

    def forward_mixed(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache: bool = True):
        # base_attn_backend.py:259-269 (NPU-only hook in the reference's dispatch); MIXED batches take the
        # extend kernels here, as in forward().
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache)

    # ------------------------------------------------------------------ forward
    def forward(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache: bool = True, **kwargs):
        mode = forward_batch.forward_mode
        if mode.is_idle():
            return q.new_empty(q.shape[0], layer.tp_q_head_num * layer.v_head_dim)
        if self._roctx:  # option `roctx` (read when the metadata is built): one named range per layer around its launches
            from .. import lib as _lib
            _lib.range_push(f"layer{layer.layer_id}.{'decode' if mode.is_decode() else 'extend'}")
            try:
                if mode.is_decode():
                    return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)
                return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)
            finally:
                _lib.range_pop()
        if mode.is_decode():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)

    @staticmethod
    def _scales(layer):
        if layer.k_scale is not None and layer.v_scale is not None:
            return layer.k_scale_float, layer.v_scale_float
        return 1.0, 1.0

    def _fused_store_ok(self, layer, k, v) -> bool:
        """rx_decode_params.k_new: the new token's rows are read from k / v and written to their slots by the decode
        kernel itself -- one of OUR pools (or a foreign one that declares ``supports_fused_decode_store``: the fused
        store never calls its set_kv_buffer), a 16-bit pool of k's dtype, D = 64 / 128, at most 16 q heads per kv head, the plain or
        shared-prefix path (not DCP / sliding-window / scaled-store), 16-byte aligned rows."""
        if self._no_fused_store or self.dcp is not None or self._is_mla_pool or not self._pool_allows_fused_store:
            return False
        pool = self.token_to_kv_pool
        if getattr(pool, "is_fp8", False) or k.dtype != pool.dtype or v.dtype != pool.dtype:
            return False
        if layer.k_scale is not None or layer.v_scale is not None:
            return False
        if layer.sliding_window_size is not None and layer.sliding_window_size > -1:
            return False
        d = layer.qk_head_dim
        if d != layer.v_head_dim or d not in (64, 128) or layer.tp_q_head_num > 16 * layer.tp_k_head_num:
            return False
        return (k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0 and k.stride(-1) == 1 and v.stride(-1) == 1
                and (layer.tp_k_head_num * d) % 8 == 0)

    def forward_decode(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache=True, sinks=None,
                       score_mod=None, aux_tensors=None):
        """triton_backend.py:1714-1864.  ``score_mod`` / ``aux_tensors`` (:1723-1724, :1861-1862): the reference's
        relative_bias_score_mod runs as the kernels' built-in bias (ops.relative_bias_score_mod); such a call stores the
        step's rows on their own and takes the plain per-request launch.  PRECONDITION of the fused store (`_fused_store_ok`): the step's slot is
        already in the page table, ``req_to_token[req_pool_indices, seq_lens - 1] == out_cache_loc`` -- what
        alloc_for_decode leaves behind (allocation.py:578-580) -- because the kernel writes the new row to the slot it
        READS for position seq_len - 1; the separate store (every foreign pool, fp8 / scaled / windowed layers) goes
        by out_cache_loc like the reference.  RX_DEBUG_CHECKS=1 asserts the equality (a host sync) before fusing."""
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        if layer.qk_head_dim != layer.v_head_dim:
            o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        else:
            o = torch.empty_like(q)
        # the KV store of the step rides in the decode launch when the kernel can take it (see _fused_store_ok)
        fuse = (save_kv_cache and k is not None and not getattr(self.forward_metadata, "draft", False)
                and score_mod is None and self._fused_store_ok(layer, k, v))  # (draft rows are top-k branches: their slot is not req_to_token[req, len - 1])
        if (fuse and self._debug_checks and forward_batch.out_cache_loc is not None
                and not torch.cuda.is_current_stream_capturing()):  # (a host sync: not inside a graph capture)
            slot = self.req_to_token[forward_batch.req_pool_indices.long(), forward_batch.seq_lens.long() - 1]
            if not torch.equal(slot.long(), forward_batch.out_cache_loc.long().view(-1)):
                raise AssertionError("fused decode store: req_to_token[req, seq_len - 1] != out_cache_loc "
                                     "(write_cache_indices / alloc_for_decode must run before the forward)")
        if save_kv_cache and k is not None and not fuse:
            if self.dcp is not None:
                self._dcp_store(layer, forward_batch, k, v)
            else:
                self.token_to_kv_pool.set_kv_buffer(layer, self._loc_info(forward_batch.out_cache_loc), k, v,
                                                    layer.k_scale, layer.v_scale)
        md = self.forward_metadata
        q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
        o3 = o.view(-1, layer.tp_q_head_num, layer.v_head_dim)
        if self.dcp is not None:
            if score_mod is not None:
                raise NotImplementedError("DCP decode does not support score_mod (as the reference, triton_backend.py:1798-1801)")
            self._forward_decode_dcp(q3, o3, layer, forward_batch, sinks)
            return o
        if score_mod is not None:
            k_descale, v_descale = self._scales(layer)
            k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
            lay = ops.kv_layout_hnd(k_buf, v_buf) if getattr(self.token_to_kv_pool, "use_hnd", False) else None
            if sinks is not None and sinks.dtype != torch.float32:
                sinks = sinks.float()
            splits = md.max_kv_splits if md.attn_logits is not None else 1
            common = dict(logit_cap=layer.logit_cap, sinks=sinks, page_size=self.page_size, kv_layout=lay,
                          xai_temperature_len=max(0, int(getattr(layer, "xai_temperature_len", -1) or 0)),
                          score_mod=score_mod, aux_tensors=aux_tensors)
            if (layer.sliding_window_size is not None and layer.sliding_window_size > -1
                    and md.window_kv_indptr is not None):
                # a sliding-window layer with score_mod (Inkling's local layers) reads the WINDOW list and its own
                # split schedule (triton_backend.py:1770-1781); rel = (len - 1) - n is then over the window list
                ops.decode_attention_fwd(q3, k_buf, v_buf, o3, md.window_kv_indptr, md.window_kv_indices, md.attn_logits,
                                         md.attn_lse, md.window_num_kv_splits, splits, layer.scaling, k_descale, v_descale,
                                         **common)
            elif md.kv_indices is not None:
                ops.decode_attention_fwd(q3, k_buf, v_buf, o3, md.kv_indptr, md.kv_indices, md.attn_logits, md.attn_lse,
                                         md.num_kv_splits, splits, layer.scaling, k_descale, v_descale, **common)
            else:
                ops.decode_attention_fwd_paged(q3, k_buf, v_buf, o3, self.req_to_token, forward_batch.req_pool_indices,
                                               forward_batch.seq_lens, md.attn_logits, md.attn_lse, md.num_kv_splits,
                                               splits, layer.scaling, k_descale, v_descale, **common)
            return o
        if self._cascade_on and not (getattr(layer, "xai_temperature_len", -1) or 0) > 0:
            k_descale, v_descale = self._scales(layer)
            k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
            hnd = getattr(self.token_to_kv_pool, "use_hnd", False)
            kn = k.view(-1, layer.tp_k_head_num, layer.qk_head_dim) if fuse else None
            vn = v.view(-1, layer.tp_v_head_num, layer.v_head_dim) if fuse else None
            self._cascade_call(q3, k_buf, v_buf, o3, layer.scaling, k_descale, v_descale, layer.logit_cap,
                          sinks if sinks is None or sinks.dtype == torch.float32 else sinks.float(),
                          page_size=self.page_size, kv_layout=ops.kv_layout_hnd(k_buf, v_buf) if hnd else None,
                          k_new=kn, v_new=vn)
            return o
        ln = self._decode_launchers.get(layer.layer_id)
        if ln is None:
            k_descale, v_descale = self._scales(layer)
            k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
            hnd = getattr(self.token_to_kv_pool, "use_hnd", False)
            ln = self._decode_launchers[layer.layer_id] = ops.DecodeLauncher(
                k_buf, v_buf, self.page_size, layer.tp_q_head_num,
                k_buf.shape[1] if hnd else k_buf.shape[-2], layer.qk_head_dim, layer.v_head_dim,
                layer.scaling, k_descale, v_descale, layer.logit_cap,
                kv_layout=ops.kv_layout_hnd(k_buf, v_buf) if hnd else None, q_dtype=q.dtype)
            ln.p.xai_temperature_len = max(0, int(getattr(layer, "xai_temperature_len", -1) or 0))
        swa = (layer.sliding_window_size is not None and layer.sliding_window_size > -1
               and md.window_kv_indptr is not None)
        if ln.version != self._md_version:
            if swa:  # sliding-window layer: the window's own indices and split schedule (triton_backend.py:1770-1781)
                ln.set_metadata(self._md_version, q3.shape[0], kv_indptr=md.window_kv_indptr,
                                kv_indices=md.window_kv_indices, num_kv_splits=md.window_num_kv_splits,
                                max_kv_splits=md.max_kv_splits if md.attn_logits is not None else 1,
                                attn_logits=md.attn_logits, attn_lse=md.attn_lse,
                                merge_counters=self._merge_counters)
            elif self.decode_index_mode == "indices" or md.draft:
                # (a draft step has one row per top-k BRANCH; the counters were sized pool * topk * heads at init.  Nothing is
                # (re)allocated on the forward path -- a captured graph holds the old pointer: a batch with more rows than
                # that takes stage 2 as its own launch for this call instead)
                mc = self._merge_counters
                if mc is not None and mc.numel() < q3.shape[0] * layer.tp_q_head_num:
                    mc = None
                ln.set_metadata(self._md_version, q3.shape[0], kv_indptr=md.kv_indptr,
                                kv_indices=md.kv_indices, num_kv_splits=md.num_kv_splits,
                                max_kv_splits=md.max_kv_splits, attn_logits=md.attn_logits,
                                attn_lse=md.attn_lse, merge_counters=mc,
                                request_order=md.request_order, partial_pairs_hint=md.partial_pairs_hint,
                                split_items=md.split_items)
            else:
                ln.set_metadata(self._md_version, q3.shape[0], req_to_token=self.req_to_token,
                                req_pool_indices=forward_batch.req_pool_indices,
                                seq_lens=forward_batch.seq_lens, num_kv_splits=md.num_kv_splits,
                                max_kv_splits=md.max_kv_splits, attn_logits=md.attn_logits,
                                attn_lse=md.attn_lse, merge_counters=self._merge_counters,
                                request_order=md.request_order, partial_pairs_hint=md.partial_pairs_hint,
                                split_items=md.split_items, units=md.decode_units)
        if sinks is not None and sinks.dtype != torch.float32:
            sinks = sinks.float()
        if fuse:
            ln(q3, o3, torch.cuda.current_stream(q.device).cuda_stream, sinks,
               k_new=k.view(-1, layer.tp_k_head_num, layer.qk_head_dim), v_new=v.view(-1, layer.tp_v_head_num, layer.v_head_dim))
        else:
            ln(q3, o3, torch.cuda.current_stream(q.device).cuda_stream, sinks)
        return o

    def forward_extend(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache=True, sinks=None,
                       score_mod=None, aux_tensors=None):
        """triton_backend.py:1250-1437 (score_mod / aux_tensors: :1259-1260, :1348-1349, :1434-1435)."""
        if layer.qk_head_dim != layer.v_head_dim:
            o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        else:
            o = torch.empty_like(q)
        if k is None or v is None:
            raise ValueError("forward_extend needs the new tokens' k and v")
        if getattr(layer, "logit_capping_method", "tanh") != "tanh":  # logit_capping_mod (triton_backend.py:83-88)
            raise ValueError(f"logit_capping_method {layer.logit_capping_method!r}: only 'tanh'")
        # (:1318-1327) cross attention, encoder-only layers and -- where the runner guarantees whole, unpadded prefills --
        # bidirectional decoder layers (image tokens) attend without the causal triangle
        causal = not (layer.is_cross_attention or layer.attn_type.value == "encoder_only"
                      or (layer.attn_type.value == "decoder_bidirectional" and self.allow_bidirectional_attention_in_extend))
        if self.dcp is not None:
            if score_mod is not None:
                raise NotImplementedError("DCP extend does not support score_mod (as the reference, triton_backend.py:1330-1333)")
            # attention first: the new tokens' K/V are read from k / v, the cache holds the prefix only
            self._forward_extend_dcp(q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
                                     k.view(-1, layer.tp_k_head_num, layer.qk_head_dim),
                                     v.view(-1, layer.tp_v_head_num, layer.v_head_dim),
                                     o.view(-1, layer.tp_q_head_num, layer.v_head_dim), layer, forward_batch, causal, sinks)
            if save_kv_cache:
                self._dcp_store(layer, forward_batch, k, v)
            return o
        if save_kv_cache:
            self.token_to_kv_pool.set_kv_buffer(layer, self._loc_info(forward_batch.out_cache_loc), k, v,
                                                layer.k_scale, layer.v_scale)
        if self.enable_deterministic:  # triton_backend.py:1339-1350
            return self._forward_extend_unified(q, o, layer, forward_batch, causal, sinks, score_mod, aux_tensors)
        md = self.forward_metadata
        k_descale, v_descale = self._scales(layer)
        k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        lay = ops.kv_layout_hnd(k_buf, v_buf) if getattr(self.token_to_kv_pool, "use_hnd", False) else None
        # sliding-window layers read the window indices (triton_backend.py:1353-1365)
        if (layer.sliding_window_size is not None and layer.sliding_window_size > -1
                and md.window_kv_indptr is not None):
            window = layer.sliding_window_size
            kv_indptr, kv_indices, window_kv_offsets = md.window_kv_indptr, md.window_kv_indices, md.window_kv_offsets
        else:
            window = layer.sliding_window_size if (layer.sliding_window_size is not None
                                                   and layer.sliding_window_size > -1) else -1
            kv_indptr, kv_indices, window_kv_offsets = md.kv_indptr, md.kv_indices, None
        page_size = self.page_size
        if (self.mla_v_is_latent_prefix and self._is_mla_pool and layer.qk_head_dim > layer.v_head_dim
                and k.shape[-1] == layer.tp_k_head_num * layer.qk_head_dim):
            v = k.view(-1, layer.tp_k_head_num, layer.qk_head_dim)[..., : layer.v_head_dim]
        if (self._is_mla_pool and ops._is_fp8_pool(k_buf) and kv_indices is not None and kv_indices.numel() > 0
                and layer.qk_head_dim == 576):
            # fp8 latent rows under an extend: the cached rows this batch reads are upcast (exactly) into a dense
            # 16-bit copy first (rx_get_mla_kv: ~1.7 KB of traffic per row against ~2 MFLOP of attention per row and
            # query) and the 16-bit MFMA kernel runs on that copy; k_scale / v_scale apply as they do on the pool
            nope, rope = self.token_to_kv_pool.get_mla_kv_buffer(layer, kv_indices, dst_dtype=q.dtype)
            k_buf = torch.cat([nope, rope], dim=-1)
            v_buf = k_buf[..., : layer.v_head_dim]
            kv_indices = torch.arange(k_buf.shape[0], dtype=torch.int64, device=k_buf.device)
            lay, page_size = None, 1
        split = ((self._verify_split_on and forward_batch.forward_mode.is_target_verify())
                 or (self._extend_split_on and forward_batch.forward_mode.is_extend() and causal
                     and not forward_batch.forward_mode.is_target_verify() and md.custom_mask is None))
        if (split and sinks is None and score_mod is None
                and (layer.qk_head_dim, layer.v_head_dim) == (self._verify_split.d, self._verify_split.dv)
                and not (getattr(layer, "xai_temperature_len", -1) or 0) > 0
                and not (layer.sliding_window_size is not None and layer.sliding_window_size > -1)
                and not (layer.logit_cap and layer.logit_cap > 0 and layer.qk_head_dim != 128)
                and layer.tp_q_head_num == self.num_head):
            dk_, dv_ = layer.qk_head_dim, layer.v_head_dim
            self._verify_split(q.view(-1, layer.tp_q_head_num, dk_), k.view(-1, layer.tp_k_head_num, dk_),
                               v.view(-1, layer.tp_v_head_num, dv_), o.view(-1, layer.tp_q_head_num, dv_), k_buf, v_buf,
                               k_descale, v_descale, sm_scale=layer.scaling, logit_cap=layer.logit_cap,
                               page_size=self.page_size, kv_layout=lay)
            return o
        # few new tokens per request (speculative verify / draft extend, short chunks): GQA-packed query rows -- the
        # G q heads of a kv head share one pass over the request's K/V (3.1-3.4x at 4-16 draft tokens over 4-8k)
        # (only when the per-head launch would over-subscribe the chip: packing trades workgroups for work, and a tiny
        # batch needs the parallelism more -- 1 request x 32k + 64 tokens: per head 636 us, packed 993 us)
        n_req = md.qo_indptr.shape[0] - 1
        packed = (md.max_extend_len is not None and md.max_extend_len <= 64 and sinks is None and score_mod is None
                  and layer.tp_q_head_num > layer.tp_k_head_num and layer.qk_head_dim == 128 == layer.v_head_dim
                  and n_req * layer.tp_q_head_num >= 2 * self.device_core_count)
        (ops.extend_attention_fwd_gqa_packed if packed else ops.extend_attention_fwd)(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
            k.view(-1, layer.tp_k_head_num, layer.qk_head_dim),
            v.view(-1, layer.tp_v_head_num, layer.v_head_dim),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim), k_buf, v_buf, md.qo_indptr,
            kv_indptr, kv_indices, md.custom_mask, causal, md.mask_indptr, md.max_extend_len, k_descale,
            v_descale, sm_scale=layer.scaling, logit_cap=layer.logit_cap, sliding_window_size=window,
            sinks=sinks, window_kv_offsets=window_kv_offsets if md.custom_mask is not None else None,
            xai_temperature_len=layer.xai_temperature_len, page_size=page_size, kv_layout=lay,
            score_mod=score_mod, aux_tensors=aux_tensors)
        return o

    def _forward_extend_unified(self, q, o, layer, forward_batch: ForwardBatch, causal: bool, sinks, score_mod=None,
                                aux_tensors=None):
        """_forward_extend_unified (triton_backend.py:1572-1712): the one-stage extend of deterministic inference.  The new
        tokens' K / V are in the pool already; prefix and new tokens are read through ONE kv list (built once per forward
        and kind of layer: rx_build_unified_kv_indices) whose tiles do not depend on where the prefix ends."""
        md = self.forward_metadata
        bs = forward_batch.batch_size
        swa = (layer.sliding_window_size is not None and layer.sliding_window_size > -1 and md.window_kv_indptr is not None)
        window = layer.sliding_window_size if (layer.sliding_window_size is not None and layer.sliding_window_size > -1) else -1
        if md.unified is None:
            md.unified = {}
        cache = md.unified
        if swa not in cache:
            pre_indptr, pre_indices = (md.window_kv_indptr, md.window_kv_indices) if swa else (md.kv_indptr, md.kv_indices)
            ext_lens = forward_batch.extend_seq_lens
            if ext_lens is None:  # TARGET_VERIFY: every request extends by the draft tokens (:1632-1647)
                n_draft = getattr(forward_batch.spec_info, "draft_token_num", None)
                if n_draft is None:
                    raise RuntimeError("extend_seq_lens is None but cannot infer from spec_info")
                ext_lens = torch.full((bs,), int(n_draft), dtype=torch.int32, device=self.device)
            start = forward_batch.extend_start_loc
            if start is None:
                start = torch.zeros_like(ext_lens)
                start[1:] = torch.cumsum(ext_lens[:-1], dim=0)
            cache[swa] = ops.build_unified_kv_indices(pre_indptr[: bs + 1], pre_indices, start.contiguous(), ext_lens.contiguous(),
                                                      forward_batch.out_cache_loc.contiguous(), bs,
                                                      max_tokens_per_request=self.max_context_len)
        u_indptr, u_indices, prefix_lens = cache[swa]
        k_descale, v_descale = self._scales(layer)
        k_buf, v_buf = self.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        lay = ops.kv_layout_hnd(k_buf, v_buf) if getattr(self.token_to_kv_pool, "use_hnd", False) else None
        ops.extend_attention_fwd_unified(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            k_buf, v_buf, k_descale, v_descale, md.qo_indptr, u_indptr, u_indices, prefix_lens, md.max_extend_len,
            custom_mask=md.custom_mask, mask_indptr=md.mask_indptr, sm_scale=layer.scaling, logit_cap=layer.logit_cap,
            is_causal=causal, sliding_window_size=window, sinks=sinks, xai_temperature_len=layer.xai_temperature_len,
            page_size=self.page_size, score_mod=score_mod, aux_tensors=aux_tensors, kv_layout=lay)
        return o

    def support_triton(self):
        return False


class HipRadixMultiStepDraftBackend:
    """TritonMultiStepDraftBackend (srt/layers/attention/triton_backend.py:1867-2040): the attention backends of EAGLE's
    ``speculative_num_steps - 1`` consecutive draft decode steps as one object.  Step i attends, for every request and
    each of its ``topk`` branches, to the request's cached tokens plus the i + 1 draft tokens the branch has written;
    the page tables of ALL steps come from one launch of rx_draft_decode_kv_indices (the reference's
    generate_draft_decode_kv_indices, cache_locs.py:56-141) into ``kv_indptr [steps, max_bs * topk + 1]`` and a
    ``kv_indices [steps, num_seqs * topk * max_context_len]`` buffer; step i's backend reads row i through
    ``forward_batch.spec_info.kv_indptr / kv_indices``."""

    needs_cpu_seq_lens: bool = False

    def __init__(self, model_runner, topk: int, speculative_num_steps: int, **backend_kwargs):
        self.topk = int(topk)
        self.speculative_num_steps = int(speculative_num_steps)
        max_bs = model_runner.req_to_token_pool.size * self.topk
        self.device = model_runner.device
        self.kv_indptr = torch.zeros((self.speculative_num_steps, max_bs + 1), dtype=torch.int32, device=self.device)
        self.attn_backends = [HipRadixAttnBackend(model_runner, skip_prefill=True, kv_indptr_buf=self.kv_indptr[i], topk=self.topk,
                                                  **backend_kwargs)
                              for i in range(self.speculative_num_steps - 1)]
        self.max_context_len = (self.attn_backends[0].max_context_len if self.attn_backends
                                else model_runner.model_config.context_len)
        self.req_to_token_pool = model_runner.req_to_token_pool
        self.pool_len = self.req_to_token_pool.req_to_token.shape[1]
        self.page_size = getattr(model_runner, "page_size", 1) or 1
        self.cuda_graph_kv_indices = None

    def common_template(self, forward_batch: ForwardBatch, kv_indices_buffer: Optional[torch.Tensor], call_fn):
        if kv_indices_buffer is None:
            kv_indices_buffer = self.cuda_graph_kv_indices
        num_seqs = forward_batch.batch_size
        bs = self.topk * num_seqs
        seq_lens_sum = forward_batch.seq_lens_sum
        if seq_lens_sum is None:  # only slice-clamps the preallocated buffer: an over-estimate is safe
            seq_lens_sum = num_seqs * self.max_context_len
        ops.generate_draft_decode_kv_indices(forward_batch.req_pool_indices, self.req_to_token_pool.req_to_token,
                                             forward_batch.seq_lens, kv_indices_buffer, self.kv_indptr, forward_batch.positions,
                                             self.topk, self.speculative_num_steps, self.page_size)
        if call_fn is None:
            return
        for i in range(self.speculative_num_steps - 1):
            forward_batch.spec_info.kv_indptr = self.kv_indptr[i, : bs + 1]
            forward_batch.spec_info.kv_indices = kv_indices_buffer[i][: draft_kv_indices_used_len(seq_lens_sum, self.topk, bs, i + 1)]
            call_fn(i, forward_batch)

    def init_forward_metadata(self, forward_batch: ForwardBatch):
        width = draft_kv_indices_buffer_width(forward_batch.batch_size, self.topk, self.max_context_len)
        kv_indices = torch.empty((self.speculative_num_steps, width), dtype=torch.int64, device=self.device)

        def call_fn(i, fb):
            # every step's backend keeps ITS OWN copies: the loop rewrites spec_info for the next step (:1967-1974)
            fb.spec_info.kv_indptr = fb.spec_info.kv_indptr.clone()
            fb.spec_info.kv_indices = fb.spec_info.kv_indices.clone()
            self.attn_backends[i].init_forward_metadata(fb)

        self.common_template(forward_batch, kv_indices, call_fn)

    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        width = draft_kv_indices_buffer_width(max_bs, self.topk, self.max_context_len)
        self.cuda_graph_kv_indices = torch.zeros((self.speculative_num_steps, width), dtype=torch.int64, device=self.device)
        cap = self.attn_backends[0].max_kv_splits if self.attn_backends else 1
        self.cuda_graph_num_kv_splits = torch.full((max_num_tokens,), cap, dtype=torch.int32, device=self.device)
        for i in range(self.speculative_num_steps - 1):
            self.attn_backends[i].init_cuda_graph_state(max_bs, max_num_tokens, kv_indices_buf=self.cuda_graph_kv_indices[i],
                                                        cuda_graph_num_kv_splits_buf=self.cuda_graph_num_kv_splits)

    def init_forward_metadata_out_graph(self, forward_batch: ForwardBatch, in_capture: bool = False):
        """Before capture AND before every replay (:2002-2036): the page tables of all steps are rebuilt into the
        address-stable buffers; the per-step backends point their metadata at row i (whole rows: the kernels stop at
        kv_indptr) and refill the shared split counts."""
        def call_fn(i, fb):
            self.attn_backends[i].init_forward_metadata_out_graph(fb, in_capture=in_capture)

        self.common_template(forward_batch, None, call_fn)

    def init_forward_metadata_in_graph(self, forward_batch: ForwardBatch) -> None:
        for b in self.attn_backends:
            b.init_forward_metadata_in_graph(forward_batch)
