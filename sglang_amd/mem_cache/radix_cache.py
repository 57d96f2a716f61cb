"""RadixCache: the prefix tree that makes requests share KV pages.

Public contract of the reference's RadixCache (srt/mem_cache/radix_cache.py:279-812) over the
native tree in libradix_hip.so (csrc/rx_radix.cpp): ``match_prefix`` -> MatchResult,
``insert`` -> InsertResult, ``evict``, ``inc_lock_ref`` / ``dec_lock_ref``, ``evictable_size``,
``protected_size``, ``total_size``, ``cache_finished_req`` / ``cache_unfinished_req``.
Token ids and KV slot ids cross the C ABI as int64 host arrays; matched slot runs come back as
one int64 tensor on the allocator's device (one H2D copy per match instead of a torch.cat of
per-node device tensors).
"""
from __future__ import annotations

import ctypes as C
from array import array
from dataclasses import dataclass
from typing import Any, List, NamedTuple, Optional, Sequence

import numpy as np
import torch

from .. import lib as _L

_POLICIES = {"lru": 0, "lfu": 1, "fifo": 2, "mru": 3, "filo": 4, "priority": 5, "slru": 6}


class RadixKey:
    """Token ids (+ optional namespace ``extra_key``), radix_cache.py:59-219 without the bigram
    (EAGLE) view."""

    __slots__ = ("token_ids", "extra_key")

    def __init__(self, token_ids: Sequence[int], extra_key: Optional[str] = None):
        self.token_ids = token_ids
        self.extra_key = extra_key

    def __len__(self):
        return len(self.token_ids)

    def __getitem__(self, idx):
        if isinstance(idx, int):
            idx = slice(idx, idx + 1)
        return RadixKey(self.token_ids[idx], self.extra_key)

    def page_aligned(self, page_size: int) -> "RadixKey":
        if page_size == 1:
            return self
        return self[: len(self) // page_size * page_size]

    def as_int64(self) -> np.ndarray:
        t = self.token_ids
        if isinstance(t, np.ndarray):
            return np.ascontiguousarray(t, dtype=np.int64)
        if isinstance(t, array) and t.typecode == "q":
            return np.frombuffer(t, dtype=np.int64) if len(t) else np.empty(0, dtype=np.int64)
        return np.asarray(list(t), dtype=np.int64)


class TreeNode:
    """Handle of a native node (the reference hands TreeNode objects to req.last_node)."""

    __slots__ = ("_cache", "id")

    def __init__(self, cache: "RadixCache", node_id: int):
        self._cache, self.id = cache, node_id

    def _info(self):
        info = (C.c_int64 * 6)()
        if self._cache._lib.rx_radix_node_info(self._cache._h, self.id, info) != 0:
            raise KeyError(f"radix node {self.id} no longer exists (evicted)")
        return list(info)

    @property
    def parent(self) -> Optional["TreeNode"]:
        p = self._info()[0]
        return None if p < 0 else TreeNode(self._cache, p)

    @property
    def lock_ref(self) -> int:
        return self._info()[2]

    @property
    def hit_count(self) -> int:
        return self._info()[3]

    @property
    def num_children(self) -> int:
        return self._info()[4]

    @property
    def priority(self) -> int:
        return self._info()[5]

    def key_len(self) -> int:
        return self._info()[1]

    def __eq__(self, other):
        return isinstance(other, TreeNode) and other.id == self.id and other._cache is self._cache

    def __hash__(self):
        return hash(self.id)

    def __repr__(self):
        return f"TreeNode(id={self.id})"


@dataclass
class MatchPrefixParams:
    key: RadixKey


class MatchResult(NamedTuple):
    device_indices: torch.Tensor
    last_device_node: Any
    last_host_node: Any
    best_match_node: Any = None
    host_hit_length: int = 0


@dataclass
class InsertParams:
    key: RadixKey
    value: Optional[torch.Tensor] = None
    priority: int = 0
    chunked: bool = False


@dataclass
class InsertResult:
    prefix_len: int
    last_device_node: Any = None


@dataclass
class EvictParams:
    num_tokens: int


@dataclass
class EvictResult:
    num_tokens_evicted: int = 0


@dataclass
class IncLockRefResult:
    delta: int


@dataclass
class DecLockRefResult:
    delta: int


@dataclass
class Req:
    """The fields of managers/schedule_batch.Req that the cache touches."""

    origin_input_ids: List[int]
    output_ids: List[int]
    req_pool_idx: Optional[int] = None
    extra_key: Optional[str] = None
    priority: int = 0
    prefix_indices: Optional[torch.Tensor] = None
    last_node: Optional[TreeNode] = None
    cache_protected_len: int = 0
    fill_ids: Optional[List[int]] = None

    def get_fill_ids(self):
        return self.fill_ids if self.fill_ids is not None else self.origin_input_ids + self.output_ids


class RadixCache:
    def __init__(self, req_to_token_pool=None, token_to_kv_pool_allocator=None, page_size: int = 1,
                 disable: bool = False, eviction_policy: str = "lru", disable_finished_insert: bool = False):
        self.disable = disable
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.page_size = page_size
        self.disable_finished_insert = disable_finished_insert
        self.eviction_policy = eviction_policy.lower()
        if self.eviction_policy not in _POLICIES:
            raise ValueError(f"unknown eviction policy {eviction_policy!r}; one of {sorted(_POLICIES)}")
        dev = getattr(token_to_kv_pool_allocator, "device", "cpu") if token_to_kv_pool_allocator else "cpu"
        self.device = torch.device(dev) if isinstance(dev, (str, torch.device)) else torch.device("cpu")
        self._lib = _L.load()
        self._h = self._lib.rx_radix_create(page_size, _POLICIES[self.eviction_policy])
        if not self._h:
            raise _L.RadixHipError("rx_radix_create failed")
        self.root_node = TreeNode(self, self._lib.rx_radix_root(self._h))
        self._empty = torch.empty((0,), dtype=torch.int64, device=self.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rx_radix_destroy(h)

    @classmethod
    def create_simulated(cls, disable: bool = False, mock_allocator=None, page_size: int = 1):
        """radix_cache.py:309-325."""
        return cls(None, mock_allocator, page_size, disable)

    # ------------------------------------------------------------------ public API
    def reset(self):
        self._lib.rx_radix_reset(self._h)
        self.root_node = TreeNode(self, self._lib.rx_radix_root(self._h))

    def match_prefix(self, params: MatchPrefixParams) -> MatchResult:
        key = params.key if isinstance(params, MatchPrefixParams) else params
        if self.disable or len(key) == 0:
            return MatchResult(self._empty, self.root_node, self.root_node, self.root_node)
        toks = key.as_int64()
        out = np.empty(len(toks), dtype=np.int64)
        last = C.c_int64(0)
        n = self._lib.rx_radix_match_prefix(
            self._h, toks.ctypes.data, len(toks),
            None if key.extra_key is None else key.extra_key.encode(), out.ctypes.data, len(out),
            C.byref(last))
        if n < 0:
            raise _L.RadixHipError("rx_radix_match_prefix failed")
        node = TreeNode(self, last.value)
        idx = torch.from_numpy(out[:n].copy()).to(self.device) if n else self._empty
        return MatchResult(idx, node, node, node)

    def insert(self, params: InsertParams) -> InsertResult:
        if self.disable:
            return InsertResult(prefix_len=0)
        key = params.key.page_aligned(self.page_size)
        toks = key.as_int64()
        if params.value is not None:
            vals = params.value[: len(key)].detach().to("cpu", torch.int64).contiguous().numpy()
        else:  # debug / test fallback: token ids as values (radix_cache.py:425-427)
            vals = toks.copy()
        last = C.c_int64(0)
        pl = self._lib.rx_radix_insert(
            self._h, toks.ctypes.data, vals.ctypes.data, len(toks),
            None if key.extra_key is None else key.extra_key.encode(), int(params.priority or 0),
            int(bool(params.chunked)), C.byref(last))
        return InsertResult(prefix_len=int(pl), last_device_node=TreeNode(self, last.value))

    def evict(self, params: EvictParams) -> EvictResult:
        if self.disable:
            return EvictResult()
        want = params.num_tokens if isinstance(params, EvictParams) else int(params)
        cap = self._lib.rx_radix_evictable_size(self._h)
        if cap == 0 or want <= 0:
            return EvictResult(0)
        slots = np.empty(cap, dtype=np.int64)
        nodes_cap = self._lib.rx_radix_num_nodes(self._h)
        seg_lens = np.empty(nodes_cap, dtype=np.int64)
        nseg = C.c_int64(0)
        n = self._lib.rx_radix_evict(self._h, want, slots.ctypes.data, cap, seg_lens.ctypes.data,
                                     nodes_cap, C.byref(nseg))
        if self.token_to_kv_pool_allocator is not None and n:
            off = 0
            dev_slots = torch.from_numpy(slots[:n].copy()).to(self.device)
            for i in range(nseg.value):  # one page-exact segment per evicted node, in heap order
                ln = int(seg_lens[i])
                self.token_to_kv_pool_allocator.free_segment(dev_slots[off: off + ln], start_pos=0)
                off += ln
        return EvictResult(num_tokens_evicted=int(n))

    def inc_lock_ref(self, node: TreeNode) -> IncLockRefResult:
        if self.disable:
            return IncLockRefResult(delta=0)
        d = self._lib.rx_radix_inc_lock_ref(self._h, node.id)
        if d == -(1 << 63):
            raise KeyError(f"radix node {node.id} does not exist")
        return IncLockRefResult(delta=int(d))

    def dec_lock_ref(self, node: TreeNode, params=None) -> DecLockRefResult:
        if self.disable:
            return DecLockRefResult(delta=0)
        d = self._lib.rx_radix_dec_lock_ref(self._h, node.id)
        if d == -(1 << 63):
            raise KeyError(f"radix node {node.id} does not exist")
        return DecLockRefResult(delta=int(d))

    def evictable_size(self):
        return int(self._lib.rx_radix_evictable_size(self._h))

    def protected_size(self):
        return int(self._lib.rx_radix_protected_size(self._h))

    def total_size(self):
        return int(self._lib.rx_radix_total_size(self._h))

    def num_nodes(self):
        return int(self._lib.rx_radix_num_nodes(self._h))

    # ------------------------------------------------------------------ request hooks
    # Both hooks are ONE call into the native tree (rx_radix_cache_req): it inserts the request's page-aligned key
    # with the slots of its req_to_token row, moves the locks and answers with the row ranges that go back to the
    # allocator (and, for a request that keeps running, the row's new cached prefix).  Python only moves the data:
    # row -> host, freed ranges -> allocator, new prefix -> row.
    _FINISHED, _INSERT, _CHUNKED = 1, 2, 4

    def _book(self, req: Req, token_ids, flags: int):
        n = len(token_ids)
        row = self.req_to_token_pool.req_to_token[req.req_pool_idx, :n].to(torch.int64)
        toks = RadixKey(token_ids, req.extra_key).as_int64()
        slots = row.cpu().numpy()
        new_slots = np.empty(n if not flags & self._FINISHED else 0, dtype=np.int64)
        out8 = (C.c_int64 * 8)()
        st = self._lib.rx_radix_cache_req(
            self._h, toks.ctypes.data, slots.ctypes.data, n,
            None if req.extra_key is None else req.extra_key.encode(), int(req.priority or 0), flags,
            int(req.cache_protected_len), -1 if req.last_node is None else req.last_node.id,
            new_slots.ctypes.data, len(new_slots), out8)
        if st != 0:
            raise _L.RadixHipError("rx_radix_cache_req failed (row shorter than the key, or a stale last_node)")
        return row, new_slots, list(out8)

    def cache_finished_req(self, req: Req, is_insert: bool = True, *, kv_len_to_handle: int):
        """radix_cache.py:434-486: the finished request's pages join the tree (or are all released), the duplicates
        and the unaligned tail go back to the allocator, its lock is dropped."""
        alloc = self.token_to_kv_pool_allocator
        if self.disable:
            lo = req.cache_protected_len
            row = self.req_to_token_pool.req_to_token[req.req_pool_idx, lo:kv_len_to_handle]
            alloc.free_segment(row.to(torch.int64), start_pos=lo)
            return
        flags = self._FINISHED | (self._INSERT if is_insert and not self.disable_finished_insert else 0)
        row, _, out = self._book(req, (req.origin_input_ids + req.output_ids)[:kv_len_to_handle], flags)
        alloc.free_segments([(row[out[0]: out[1]], out[0]), (row[out[2]: out[3]], out[2])])

    def cache_unfinished_req(self, req: Req, chunked: bool = False):
        """radix_cache.py:488-553: a request that keeps running (chunked prefill, or between decode batches) parks
        its whole pages in the tree, takes over the tree's copy of that prefix and re-locks the deeper node."""
        if self.disable:
            return
        row, new_slots, out = self._book(req, req.get_fill_ids(), self._CHUNKED if chunked else 0)
        self.token_to_kv_pool_allocator.free_segment(row[out[0]: out[1]], start_pos=out[0])
        m, lo = out[5], req.cache_protected_len
        cached = torch.from_numpy(new_slots[:m]).to(self.device) if m else self._empty
        if m > lo:  # the prefix may now be pages another request cached first
            self.req_to_token_pool.req_to_token[req.req_pool_idx, lo:m] = cached[lo:].to(torch.int32)
        req.cache_protected_len = m
        req.last_node = TreeNode(self, out[6])
        req.prefix_indices = cached if m == row.numel() else torch.cat([cached, row[m:]])


def plan_shared_prefix_groups(last_nodes, seq_lens=None, min_shared: int = 1024, min_members: int = 2):
    """Shared-prefix groups of a decode batch from the radix tree (the planner of ops.CascadeGroups).

    ``last_nodes[i]``: the node request i's cached prefix ends in -- ``MatchResult.last_device_node`` of
    RadixCache.match_prefix (srt/mem_cache/radix_cache.py:352-430), kept as ``req.last_node`` by the scheduler.  Every
    ancestor n of it is a prefix of ``depth(n)`` tokens (the key lengths root -> n summed) that request i shares, slot
    for slot, with every other request below n.  Reading the prefix of n once for its ``count(n)`` requests saves
    ``(count(n) - 1) * depth(n)`` K/V row reads; a request joins ONE group (single-level cascade), so the groups are an
    antichain of the tree, and the best one is a bottom-up choice per node: the node itself, or the best of its children.

    Returns [(member batch rows (ascending), shared token count)], largest saving first.  ``seq_lens`` (host ints,
    optional) caps the shared count at the shortest member (a prefix node cannot be longer, this only guards a caller
    that truncated a request).  Works on this package's TreeNode handles and on the reference's TreeNode objects
    (``.parent``, ``.key``)."""
    def ident(n):
        return getattr(n, "id", None) if getattr(n, "id", None) is not None else id(n)

    def klen(n):
        f = getattr(n, "key_len", None)
        if callable(f):
            return int(f())
        k = getattr(n, "key", None)
        return 0 if k is None else len(k)

    info = {}      # node id -> [parent id or None, key length, requests below, depth]
    for i, node in enumerate(last_nodes):
        n, path = node, []
        while n is not None:
            k = ident(n)
            ent = info.get(k)
            if ent is None:
                par = n.parent
                ent = info[k] = [None if par is None else ident(par), klen(n) if par is not None else 0, [], None]
                path.append(k)
                n = par
            else:
                path.append(k)
                n = None
        # every node on the whole path (also the part found already) counts request i
        k = ident(node)
        while k is not None:
            info[k][2].append(i)
            k = info[k][0]
    def depth(k):  # iterative: a long radix chain must not hit the recursion limit
        chain = []
        while k is not None and info[k][3] is None:
            chain.append(k)
            k = info[k][0]
        d = info[k][3] if k is not None else 0
        for c in reversed(chain):
            d += info[c][1]
            info[c][3] = d
        return d if chain else (info[k][3] if k is not None else 0)

    children = {}
    for k, ent in info.items():
        children.setdefault(ent[0], []).append(k)
    order = sorted(info, key=depth, reverse=True)  # children before parents
    best = {}      # node -> (saving, [group nodes])
    for k in order:
        cnt, d = len(info[k][2]), depth(k)
        own = (cnt - 1) * d if (cnt >= min_members and d >= min_shared) else 0
        sub_gain, sub_nodes = 0, []
        for c in children.get(k, []):
            g, nodes = best[c]
            sub_gain += g
            sub_nodes += nodes
        best[k] = (own, [k]) if own > sub_gain else (sub_gain, sub_nodes)
    roots = children.get(None, [])
    groups = []
    for r in roots:
        for k in best[r][1] if best[r][0] > 0 else []:
            members = sorted(info[k][2])
            L = depth(k)
            if seq_lens is not None:
                # at most seq_len - 1: with L == seq_len a member's suffix is empty -- the decode launch then never
                # walks (or, fused, stores) the step's newest row
                L = min(L, min(int(seq_lens[i]) for i in members) - 1)
            if L >= min_shared and len(members) >= min_members:
                groups.append((members, int(L)))
    groups.sort(key=lambda g: -(len(g[0]) - 1) * g[1])
    return groups
