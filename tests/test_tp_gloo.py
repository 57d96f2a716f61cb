"""Multi-process CPU tests (gloo, world_size 2) of the tensor-parallel path: head sharding,
row-parallel o_proj and the sum all-reduce reproduce the unsharded attention block.  The
attention of each shard is produced by the oracle here (the product computes it on the GPU only);
what is under test is sglang_amd.parallel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import radix_oracle as orc
from sglang_amd.parallel import shard_heads


def test_shard_heads_matches_llama_rules():
    # Llama-3-8B / 70B: Hq 32/64, Hkv 8 (models/llama.py:158-171)
    s = shard_heads(32, 8, 2, 1)
    assert (s.num_q_heads, s.num_kv_heads, s.q_head_start, s.kv_head_start, s.kv_replicas) == (16, 4, 16, 4, 1)
    s = shard_heads(32, 8, 8, 5)
    assert (s.num_q_heads, s.num_kv_heads, s.q_head_start, s.kv_head_start) == (4, 1, 20, 5)
    s = shard_heads(64, 8, 16, 5)  # kv heads replicated: 2 ranks per kv head
    assert (s.num_q_heads, s.num_kv_heads, s.kv_head_start, s.kv_replicas) == (4, 1, 2, 2)
    with pytest.raises(ValueError):
        shard_heads(32, 8, 3, 0)
    with pytest.raises(ValueError):
        shard_heads(32, 8, 12, 0)
    # the q heads of a rank all map onto the kv heads it owns
    for tp in (1, 2, 4, 8, 16):
        for rk in range(tp):
            s = shard_heads(64, 8, tp, rk)
            for hq in range(s.q_head_start, s.q_head_start + s.num_q_heads):
                kvh = hq // (64 // 8)
                assert s.kv_head_start <= kvh < s.kv_head_start + s.num_kv_heads


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sglang_amd.parallel import RowParallelOProj, TPGroup, shard_heads

        rng = np.random.default_rng(0)  # same data on every rank (the scheduler state is replicated)
        bs, HQ, HKV, D, HID = 3, 8, 2, 16, 32
        lens = np.array([5, 9, 2])
        pool = 32
        kb = rng.standard_normal((pool, HKV, D)).astype(np.float32)
        vb = rng.standard_normal((pool, HKV, D)).astype(np.float32)
        q = rng.standard_normal((bs, HQ, D)).astype(np.float32)
        w_o = rng.standard_normal((HQ * D, HID)).astype(np.float32)
        r2t = np.zeros((bs + 1, 16), dtype=np.int32)
        perm = rng.permutation(pool - 1) + 1
        o = 0
        for i, n in enumerate(lens):
            r2t[i + 1, :n] = perm[o:o + n]; o += n
        rpi = np.arange(1, bs + 1)
        kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
        full = orc.decode_attention(q, kb, vb, kv_indptr, kv_indices, D ** -0.5)
        want = full.reshape(bs, HQ * D) @ w_o.astype(np.float64)

        group = TPGroup()
        assert group.world_size == world and group.rank == rank
        sh = shard_heads(HQ, HKV, world, rank)
        qs = q[:, sh.q_head_start: sh.q_head_start + sh.num_q_heads]
        ks = kb[:, sh.kv_head_start: sh.kv_head_start + sh.num_kv_heads]
        vs = vb[:, sh.kv_head_start: sh.kv_head_start + sh.num_kv_heads]
        part = orc.decode_attention(qs, ks, vs, kv_indptr, kv_indices, D ** -0.5)  # same page table
        proj = RowParallelOProj(torch.from_numpy(w_o).double(), sh, D, group)
        y = proj.forward(torch.from_numpy(part.reshape(bs, -1)), overlap=(rank == 0)).wait()
        np.testing.assert_allclose(y.numpy(), want, atol=1e-9)
        # deterministic inference: the group built from server_args carries the flag; on CPU
        # tensors (this test) the reduce stays on the group's backend, a GPU reduce would demand the peer-to-peer context
        class SA:
            enable_deterministic_inference = True

        gdet = TPGroup.from_server_args(None, SA, torch.device("cpu"))
        assert gdet.deterministic and gdet.custom_ar is None
        yd = RowParallelOProj(torch.from_numpy(w_o).double(), sh, D, gdet).forward(torch.from_numpy(part.reshape(bs, -1))).wait()
        np.testing.assert_allclose(yd.numpy(), want, atol=1e-9)
        # the quick all-reduce (C3) under its environment switch: on a CPU group the communicator stays disabled (no region,
        # no kernel) and every reduce goes to the group's backend; its level parses as the reference's does
        os.environ["ROCM_QUICK_REDUCE_QUANTIZATION"] = "INT6"
        try:
            from sglang_amd.parallel import QuickAllReduce, QuickReduceRegime
            qr = QuickAllReduce(None, "cpu")
            assert qr.disabled and qr.qr_quant_level is QuickReduceRegime.INT6 and qr.use_fp16_kernels == 1
            gq = TPGroup(None, quick_ar=qr)
            assert gq.quick_ar is None
            big = torch.full((1 << 20,), float(rank + 1), dtype=torch.bfloat16)   # 2 MiB: inside the level's window on a GPU
            gq.all_reduce(big)
            assert float(big[0]) == sum(range(1, world + 1))
            gcpu = TPGroup.from_server_args(None, None, torch.device("cpu"))
            assert gcpu.quick_ar is None and not gcpu.deterministic
        finally:
            del os.environ["ROCM_QUICK_REDUCE_QUANTIZATION"]
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_tp2_row_parallel_allreduce_matches_unsharded(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _dcp_worker(rank, world, port, out_dir):
    """Decode context parallel on CPU: the three exchanges of sglang_amd.attention.dcp.DcpGroup (all-gather of the q
    heads, of the LSEs, the all-reduce) carry ORACLE partials -- every rank attends the tokens it owns (position %
    world == rank) with the gathered heads -- and the joined result must be the attention over the whole sequence."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sglang_amd.attention.dcp import DcpGroup, get_dcp_lens

        grp = DcpGroup(world, rank)
        rng = np.random.default_rng(1)  # same data on every rank
        bs, H_ALL, D = 3, 4, 16
        hl = H_ALL // world
        lens = np.array([7, 1, 12])
        pool = 40
        kb = rng.standard_normal((pool, 1, D)).astype(np.float32)       # virtual cache, one kv head
        vb = rng.standard_normal((pool, 1, D)).astype(np.float32)
        q = rng.standard_normal((bs, H_ALL, D)).astype(np.float32)
        r2t = np.zeros((bs + 1, 16), dtype=np.int32)
        perm = rng.permutation(pool - 1) + 1
        o = 0
        for i, n in enumerate(lens):
            r2t[i + 1, :n] = perm[o:o + n]; o += n
        rpi = np.arange(1, bs + 1)
        kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
        want = orc.decode_attention(q, kb, vb, kv_indptr, kv_indices, D ** -0.5)
        # 1. gather the q heads: rank r contributes heads [r * hl, (r + 1) * hl)
        q_all = grp.all_gather_heads(torch.from_numpy(q[:, rank * hl:(rank + 1) * hl].copy()))
        assert torch.equal(q_all, torch.from_numpy(q))
        # 2. local attention over the tokens this rank owns (the oracle here; rx_decode_attn on the GPU)
        assert np.array_equal(get_dcp_lens(torch.from_numpy(lens), world, rank).numpy(), orc.dcp_lens(lens, world, rank))
        o_loc = np.zeros((bs, H_ALL, D))
        lse_loc = np.full((bs, H_ALL), -np.inf)
        for b in range(bs):
            mine = [int(r2t[b + 1, p]) for p in range(lens[b]) if p % world == rank]
            if not mine:
                continue
            s = np.einsum("hd,td->ht", q[b].astype(np.float64), kb[mine, 0].astype(np.float64)) * D ** -0.5
            m = s.max(-1, keepdims=True)
            p = np.exp(s - m)
            o_loc[b] = (p / p.sum(-1, keepdims=True)) @ vb[mine, 0].astype(np.float64)
            lse_loc[b] = (m + np.log(p.sum(-1, keepdims=True)))[:, 0]
        # 3. cp_lse_ag_out_rs_mha: all-gather the LSEs, scale, all-reduce, keep the local heads
        lses = grp.all_gather_lse(torch.from_numpy(lse_loc))
        assert lses.shape == (world, bs, H_ALL) and torch.equal(lses[rank], torch.from_numpy(lse_loc))
        scaled, _, _ = orc.dcp_merge(np.stack([o_loc] * world), lses.numpy())   # row `rank` is this rank's contribution
        total = grp.all_reduce(torch.from_numpy(scaled[rank].copy()))
        np.testing.assert_allclose(total.numpy()[:, rank * hl:(rank + 1) * hl], want[:, rank * hl:(rank + 1) * hl],
                                   atol=1e-9)
        open(os.path.join(out_dir, f"dcp_ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_dcp2_exchanges_reproduce_whole_sequence_attention(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_dcp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"dcp_ok{r}") for r in range(world))


def test_deterministic_collectives_switch_follows_the_reference_precedence(monkeypatch):
    """GroupCoordinator._deterministic_collectives_enabled (parallel_state.py:1204-1208) / _use_amd_deterministic_impl
    (custom_all_reduce.py:415-421): SGLANG_USE_1STAGE_ALLREDUCE wins when set, else SGLANG_ENABLE_DETERMINISTIC_INFERENCE
    or --enable-deterministic-inference; a world-1 group never turns the mode on."""
    from sglang_amd.parallel import TPGroup

    class On:
        enable_deterministic_inference = True

    class Off:
        enable_deterministic_inference = False

    monkeypatch.delenv("SGLANG_USE_1STAGE_ALLREDUCE", raising=False)
    monkeypatch.delenv("SGLANG_ENABLE_DETERMINISTIC_INFERENCE", raising=False)
    f = TPGroup.deterministic_collectives_enabled
    assert f(On) and not f(Off) and not f(None)
    monkeypatch.setenv("SGLANG_ENABLE_DETERMINISTIC_INFERENCE", "1")
    assert f(Off)
    monkeypatch.setenv("SGLANG_USE_1STAGE_ALLREDUCE", "0")
    assert not f(On)
    monkeypatch.setenv("SGLANG_USE_1STAGE_ALLREDUCE", "true")
    assert f(Off)
    assert not TPGroup(None, deterministic=True).deterministic   # world size 1: nothing to reduce
