"""Dispatch coverage of rx_extend_attn / rx_decode_attn (round 4; VERDICT r03 "what's weak" 2: a wrong-answer bug lived
a round behind a green suite because no test could tell WHICH kernel instance a call had taken).

* CPU: every attention-kernel template instance in libradix_hip.so's symbol table (host launch stubs, demangled) must be
  the `expect` of a case of the matrix below -- a new instance, or a dispatch branch nobody exercises, fails here.
* GPU: every case runs through the C ABI, asserts that rx_last_dispatch() names exactly that instance, and holds the
  output to parity_util.check_out against the fp64 oracle (extend_attention_fwd, kernels/ops/attention/
  extend_attention.py:664-812; decode_attention_fwd, decode_attention.py:968-1044).  Run-time-only dispatch facts that
  hid the round-3 bug -- the GQA group size of the packed-row kernels, page layouts -- cycle through the cases."""
import itertools
import os
import re
import subprocess
import zlib

import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = ("extend_mfma32_kernel", "extend_mfma32_uni_kernel", "extend_mfma64_kernel", "extend_mfma_kernel", "extend_generic_kernel", "extend_d256_kernel", "extend_nd_kernel",
            "extend_mla_kernel", "decode_mfma_kernel", "decode_mfma_bias_kernel", "decode_generic_kernel", "decode_mla_kernel", "decode_mla8_dma_kernel", "decode_mla8_t64_kernel")
TN = {"bf16": "rx::BF16", "f16": "rx::F16"}
TB = {True: "true", False: "false"}
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
GROUPS = [(4, 4), (4, 2), (8, 2), (8, 1), (16, 1)]  # (Hq, Hkv): group 1, 2, 4, 8, 16


# ----------------------------------------------------------------------------------------------------------------------
# the case matrix: one entry per template instance, `expect` spelled as the symbol demangles
# ----------------------------------------------------------------------------------------------------------------------
def _cases():
    out = []
    n = itertools.count()

    def heads(k=None):  # cycle the GQA group sizes through the cases
        return GROUPS[(next(n) if k is None else k) % len(GROUPS)]

    # ---- extend, D = 128, 32x32x16: <T, IdxT, LINEAR, VSCALE, NW, KV8, PLAIN, PKC>
    for dt, idx, lin, nw in itertools.product(TN, ("int", "long"), (False, True), (4, 8)):
        variants = [("plain", 0), ("extras", False), ("extras", True), ("fp8", False), ("fp8", True)]
        variants += [("plain", 4), ("plain", 8)]  # (packed PLAIN rows: eight waves, or -- few tiles, round 5 -- four)
        for kind, arg in variants:
            c = dict(fam="extend", dt=dt, idx=idx, lin=lin, dk=128, dv=128, opts={"ext32_small_wg": int(nw == 4)},
                     ext=[300, 40, 257, 129], pre=[0, 70, 200, 33], kw={})
            if kind == "plain":
                pkc = arg
                if pkc:  # the packed PLAIN instances: GQA 4 / 8, causal, long extends (autopack)
                    c["hq"], c["hkv"] = (8, 2) if pkc == 4 else (8, 1)
                    c["opts"]["ext32_pack_min_wgs"] = 0  # (four requests: the packed grid is below the chip-coverage gate)
                    if nw == 4:
                        c["opts"]["ext32_pack_min_tiles"] = 0  # (no length hint: the tile estimate is below the packing gate)
                else:    # never auto-packed: group 1, 2 or 16
                    c["hq"], c["hkv"] = [(4, 4), (4, 2), (16, 1)][next(n) % 3]
                vs, kv8, plain = False, False, True
            else:
                c["hq"], c["hkv"] = heads()
                pkc, plain, kv8, vs = 0, False, kind == "fp8", bool(arg)
                c["opts"]["ext32_autopack"] = 0
                if kind == "extras":
                    c["kw"]["logit_cap"] = 30.0
                else:
                    c["fp8"] = True
                    c["kw"]["k_scale"] = 0.9
                if vs:
                    c["kw"]["v_scale"] = 1.25
            c["expect"] = (f"extend_mfma32_kernel<{TN[dt]}, {idx}, {TB[lin]}, {TB[vs]}, {nw}, {TB[kv8]}, {TB[plain]}, {pkc}>")
            out.append(c)
    # ---- extend, D = 128, the unified (deterministic) instance, round 6: <T, IdxT, LINEAR, NW>
    for dt, idx, lin, nw in itertools.product(TN, ("int", "long"), (False, True), (4, 8)):
        hq, hkv = heads()
        out.append(dict(fam="extend_unified", dt=dt, idx=idx, lin=lin, dk=128, dv=128, hq=hq, hkv=hkv, opts={"ext32_small_wg": int(nw == 4)},
                        ext=[300, 40, 257, 129], pre=[0, 70, 200, 33], kw={},
                        expect=f"extend_mfma32_uni_kernel<{TN[dt]}, {idx}, {TB[lin]}, {nw}>"))
    # ---- extend, 16x16x32 kernel of rx_extend.hip: <T, D, IdxT, LINEAR, VSCALE, PLAIN, CB>
    for dt, idx, lin in itertools.product(TN, ("int", "long"), (False, True)):
        for vs, plain, cb in [(False, True, 4), (False, True, 2), (True, True, 2), (False, False, 2), (True, False, 2)]:
            hq, hkv = heads()
            c = dict(fam="extend", dt=dt, idx=idx, lin=lin, dk=64, dv=64, hq=hq, hkv=hkv, pre=[0, 70, 200, 33],
                     ext=[520, 40, 257, 129] if cb == 4 else [100, 40, 57, 129], kw={},
                     opts={"extend_d256_at64": 0},
                     expect=f"extend_mfma_kernel<{TN[dt]}, 64, {idx}, {TB[lin]}, {TB[vs]}, {TB[plain]}, {cb}>")
            if vs:
                c["kw"]["v_scale"] = 1.25
            if not plain:
                c["kw"]["sliding_window_size"] = 90
            out.append(c)
        for vs in (False, True):
            hq, hkv = heads()
            c = dict(fam="extend", dt=dt, idx=idx, lin=lin, dk=128, dv=128, hq=hq, hkv=hkv, pre=[0, 70, 200, 33],
                     ext=[100, 40, 57, 129], kw={"v_scale": 1.25} if vs else {}, opts={"extend_16x16_d128": 1},
                     expect=f"extend_mfma_kernel<{TN[dt]}, 128, {idx}, {TB[lin]}, {TB[vs]}, false, 2>")
            out.append(c)
        hq, hkv = heads()
        out.append(dict(fam="extend", dt=dt, idx=idx, lin=lin, dk=80, dv=80, hq=hq, hkv=hkv, pre=[0, 70, 33], ext=[60, 9, 33],
                        kw={}, opts={}, expect=f"extend_generic_kernel<{TN[dt]}, {idx}, {TB[lin]}>"))
    # ---- extend, the AGPR / LDS-DMA template: <T, DK, DV, EX>  (packed rows: every group size per head-dim pair)
    for dt, (dk, dv), ex in itertools.product(TN, ((256, 256), (192, 128), (192, 192), (96, 96), (64, 64), (128, 128)), (False, True)):
        for gi in range(len(GROUPS)):
            hq, hkv = heads(gi)
            c = dict(fam="extend", dt=dt, idx=("int", "long")[gi % 2], lin=bool((gi + ex) % 2), dk=dk, dv=dv, hq=hq, hkv=hkv,
                     pre=[0, 70, 200, 33], ext=[300, 40, 257, 140], kw={"logit_cap": 30.0} if ex else {},
                     opts={"extend_d256_at128": 1} if dk == 128 else {},
                     expect=f"extend_d256_kernel<{TN[dt]}, {dk}, {dv}, {TB[ex]}>", runtime=f"g{hq // hkv}")
            out.append(c)
    # ---- extend, the 16x16x32 kernel for the other head dims: <T, DK, DV, BIG, PLAIN>
    for dt, (dk, dv), big, plain in itertools.product(TN, ((256, 256), (192, 128), (192, 192), (96, 96)), (False, True), (False, True)):
        if big and dk <= 128:
            continue
        hq, hkv = heads()
        c = dict(fam="extend", dt=dt, idx="long", lin=bool(next(n) % 2), dk=dk, dv=dv, hq=hq, hkv=hkv, pre=[0, 70, 200, 33],
                 ext=[300, 40, 257, 140] if big else [20, 9, 57, 30], kw={} if plain else {"sliding_window_size": 90},
                 opts={"extend_d256": 0, "extend_d256_at96": 0}, expect=f"extend_nd_kernel<{TN[dt]}, {dk}, {dv}, {TB[big]}, {TB[plain]}>")
        out.append(c)
    # ---- extend, latent MLA: <T, OWN_V>
    for dt, ov, lin in itertools.product(TN, (False, True), (False, True)):
        out.append(dict(fam="extend", dt=dt, idx="long", lin=lin, dk=576, dv=512, hq=(16, 5)[ov], hkv=1, pre=[0, 70, 200, 33],
                        ext=[30, 9, 57, 12], kw={}, opts={}, mla=True, own_v=ov,
                        expect=f"extend_mla_kernel<{TN[dt]}, {TB[ov]}>"))
    # ---- decode, MFMA kernel: <T, D, IdxT, LINEAR, KV8, FUSE, OCC3>
    for dt, idx, lin in itertools.product(TN, ("int", "long"), (False, True)):
        for d, kv8, fuse, occ3 in [(64, False, False, False), (64, False, True, False), (64, True, False, False),
                                   (128, False, False, False), (128, False, True, False), (128, True, False, False),
                                   (96, False, False, False), (256, False, False, False),
                                   (128, False, False, True), (128, False, True, True)]:
            hq, hkv = heads()
            out.append(dict(fam="decode", dt=dt, idx=idx, lin=lin, dk=d, dv=d, hq=hq, hkv=hkv, fp8=kv8, fuse=fuse, occ3=occ3,
                            mode=("indices" if idx == "long" else ("r2t", "indices")[next(n) % 2]),
                            expect=f"decode_mfma_kernel<{TN[dt]}, {d}, {idx}, {TB[lin]}, {TB[kv8]}, {TB[fuse]}, {TB[occ3]}>"))
        for d in (64, 128):  # the relative-position score bias (round 5): <T, D, IdxT, LINEAR>
            hq, hkv = heads()
            out.append(dict(fam="decode", dt=dt, idx=idx, lin=lin, dk=d, dv=d, hq=hq, hkv=hkv, bias=True,
                            mode=("indices" if idx == "long" else ("r2t", "indices")[next(n) % 2]),
                            expect=f"decode_mfma_bias_kernel<{TN[dt]}, {d}, {idx}, {TB[lin]}>"))
        hq, hkv = heads()
        out.append(dict(fam="decode", dt=dt, idx=idx, lin=lin, dk=80, dv=80, hq=hq, hkv=hkv, mode="indices",
                        expect=f"decode_generic_kernel<{TN[dt]}, {idx}, {TB[lin]}>"))
        for kind in ("rows16", "fp8_staged", "fp8_dma", "fp8_t64"):
            exp = (f"decode_mla8_dma_kernel<{TN[dt]}, {idx}, {TB[lin]}>" if kind == "fp8_dma"
                   else f"decode_mla8_t64_kernel<{TN[dt]}, {idx}, {TB[lin]}>" if kind == "fp8_t64"
                   else f"decode_mla_kernel<{TN[dt]}, {idx}, {TB[lin]}, {TB[kind != 'rows16']}>")
            out.append(dict(fam="decode", dt=dt, idx=idx, lin=lin, dk=576, dv=512, hq=(16, 128, 5)[next(n) % 3], hkv=1, mla=True,
                            fp8=kind != "rows16", mode="indices" if idx == "long" else "r2t",
                            opts={"decode_mla8_dma": 0} if kind == "fp8_staged" else ({"decode_mla8_t64": int(kind == "fp8_t64")} if kind in ("fp8_dma", "fp8_t64") else {}),
                            expect=exp))
    # int32 kv_indices only exist in "indices" mode; a req_to_token walk has no IdxT (the launcher takes int)
    for c in out:
        if c["fam"] == "decode" and c["idx"] == "int" and c.get("mode") == "indices":
            c["idx_dtype"] = torch.int32
    return out


CASES = _cases()


def _cid(c):
    return re.sub(r"[ :]", "", c["expect"]).replace("rx", "") + ("|" + c["runtime"] if c.get("runtime") else "")


# ----------------------------------------------------------------------------------------------------------------------
def test_every_attention_kernel_instance_has_a_parity_case():
    from sglang_amd import build

    if not os.path.exists(build.LIB_PATH):
        pytest.skip("libradix_hip.so not built")
    syms = subprocess.run(["nm", build.LIB_PATH], capture_output=True, text=True, check=True).stdout
    mangled = sorted({ln.split()[-1] for ln in syms.splitlines() if "__device_stub__" in ln})
    dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.splitlines()
    inst = set()
    for d in dem:
        m = re.search(r"__device_stub__(\w+<.*>)\(", d)
        if m and m.group(1).split("<")[0] in FAMILIES:
            inst.add(m.group(1))
    assert len(inst) > 250, f"only {len(inst)} attention-kernel instances found: symbol parsing broke?"
    covered = {c["expect"] for c in CASES}
    missing = sorted(inst - covered)
    assert not missing, f"{len(missing)} kernel instances without a dispatch-asserting parity case, e.g. {missing[:5]}"
    stale = sorted(covered - inst)
    assert not stale, f"cases that name instances the library does not have: {stale[:5]}"
    # ... and for the packed-row template every GQA group size of every head-dim pair
    for dims in ("256, 256", "192, 128", "192, 192", "96, 96", "64, 64", "128, 128"):
        got = {c["runtime"] for c in CASES if c["expect"].startswith("extend_d256_kernel") and f", {dims}," in c["expect"]}
        assert got == {"g1", "g2", "g4", "g8", "g16"}, (dims, got)


# ----------------------------------------------------------------------------------------------------------------------
def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _device_pool(ops, kb, vb, lin, ps, hnd):
    """canonical pools [slots, H, D] -> device tensors, page_size, layout, and a function giving the device K rows back
    as [slots, H, D].  linear = plain slot-major rows; otherwise HND pages [pages, H, page, D], with two pad
    tokens per (page, head) when one kv head would make plain HND slot-major."""
    if lin:
        kd, vd = kb.to(DEV), vb.to(DEV)
        return kd, vd, 1, None, lambda: kd
    pages = kb.shape[0] // ps
    pad = 0 if (hnd and kb.shape[1] > 1) else 2   # (with one kv head plain HND IS slot-major: two pad tokens per page then)
    kd = torch.zeros(pages, kb.shape[1], ps + pad, kb.shape[2], dtype=kb.dtype)
    vd = torch.zeros(pages, vb.shape[1], ps + pad, vb.shape[2], dtype=vb.dtype)
    kd[:, :, :ps], vd[:, :, :ps] = kb.view(pages, ps, *kb.shape[1:]).permute(0, 2, 1, 3), vb.view(pages, ps, *vb.shape[1:]).permute(0, 2, 1, 3)
    kd, vd = kd.to(DEV)[:, :, :ps], vd.to(DEV)[:, :, :ps]
    return kd, vd, ps, ops.kv_layout_hnd(kd, vd), lambda: kd.permute(0, 2, 1, 3).reshape(pages * ps, *kb.shape[1:])


def _run_extend(c, ops, rxlib):
    dtype = DT[c["dt"]]
    hq, hkv, dk, dv, ps = c["hq"], c["hkv"], c["dk"], c["dv"], 16
    pre, ext = np.asarray(c["pre"], np.int64), np.asarray(c["ext"], np.int64)
    bs, T = len(pre), int(ext.sum())
    rng = np.random.default_rng(zlib.crc32((c["expect"] + c.get("runtime", "")).encode()))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    n_pages = int(sum(-(-int(p) // ps) for p in pre)) + 3
    pool = n_pages * ps
    page_ids = rng.permutation(np.arange(1, n_pages))
    kvi, kvp, pi = [], [0], 0
    for p in pre:
        npg = -(-int(p) // ps)
        kvi.append((page_ids[pi: pi + npg, None] * ps + np.arange(ps)[None]).reshape(-1)[: int(p)])
        pi += npg
        kvp.append(kvp[-1] + int(p))
    kvi = np.concatenate(kvi).astype(np.int64)
    fp8 = c.get("fp8", False)
    pdt = torch.float8_e4m3fn if fp8 else dtype
    kw = dict(c["kw"])
    ks, vs = kw.pop("k_scale", 1.0), kw.pop("v_scale", 1.0)
    if c.get("mla"):
        kb = torch.randn(pool, 1, dk, generator=g).to(pdt)
        vb = kb[..., :dv]
        ke = torch.randn(T, 1, dk, generator=g).to(dtype)
        ve = ke[..., :dv].contiguous() if c["own_v"] else ke[..., :dv]
    else:
        kb = torch.randn(pool, hkv, dk, generator=g).to(pdt)
        vb = torch.randn(pool, hkv, dv, generator=g).to(pdt)
        ke = torch.randn(T, hkv, dk, generator=g).to(dtype)
        ve = torch.randn(T, hkv, dv, generator=g).to(dtype)
    q = torch.randn(T, hq, dk, generator=g).to(dtype)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    sm = dk ** -0.5
    if fp8:
        kbo, vbo = orc.fp8_e4m3fn_decode(kb.view(torch.uint8).numpy()), orc.fp8_e4m3fn_decode(vb.view(torch.uint8).numpy())
    else:
        kbo, vbo = _bits(kb), _bits(vb.contiguous())
    okw = dict(kw, k_scale=ks, v_scale=vs)
    aux = None
    if c.get("bias"):  # relative-position score bias: extent 150 puts tiles inside, across and beyond the extent (fp32 / 16-bit aux by case)
        aux = (1.5 * torch.randn(T, hq, 150, generator=g)).to(dtype if len(c["expect"]) % 2 else torch.float32)
        okw["score_bias"] = aux.double().numpy()
        kw = dict(kw, score_mod=ops.relative_bias_score_mod, aux_tensors=[aux.to(DEV)])
    # (the |V| twin of check_out's absw comes out of the same pass: return_absw)
    want, absw = orc.extend_attention(_bits(q), _bits(ke), _bits(ve.contiguous()), kbo, vbo, qo, np.asarray(kvp, np.int32), kvi,
                                      is_causal=True, sm_scale=sm, return_absw=True, **okw)
    if c.get("mla"):  # one latent tensor; v is its view (the pool always aliases) -- padded pages for the non-linear form
        if c["lin"]:
            kd = kb.to(DEV)
            vd, page, lay = kd[..., :dv], 1, None
        else:
            kd = torch.zeros(n_pages, 1, ps + 2, dk, dtype=pdt)
            kd[:, :, :ps] = kb.view(n_pages, ps, 1, dk).permute(0, 2, 1, 3)
            kd = kd.to(DEV)[:, :, :ps]
            vd, page = kd[..., :dv], ps
            lay = ops.kv_layout_hnd(kd, vd)
        ked = ke.to(DEV)
        ved = ked[..., :dv].contiguous() if c["own_v"] else ked[..., :dv]
    else:
        kd, vd, page, lay, _ = _device_pool(ops, kb, vb, c["lin"], ps, hnd=(len(c["expect"]) % 2 == 0))
        ked, ved = ke.to(DEV), ve.to(DEV)
    o = torch.full((T, hq, dv), float("nan"), dtype=dtype, device=DEV)
    idt = torch.int32 if c["idx"] == "int" else torch.int64
    hint = 4096 if c["opts"].get("ext32_small_wg", 1) == 0 else None
    opts = [rxlib.option(k, v) for k, v in c["opts"].items()]
    for cm in opts:
        cm.__enter__()
    try:
        ops.extend_attention_fwd(q.to(DEV), ked, ved, o, kd, vd, torch.from_numpy(qo).to(DEV),
                                 torch.tensor(kvp, dtype=torch.int32, device=DEV), torch.from_numpy(kvi).to(DEV).to(idt), None,
                                 True, None, int(ext.max()), ks, vs, sm_scale=sm, page_size=page, kv_layout=lay,
                                 avg_kv_len_hint=hint, **kw)
        torch.cuda.synchronize()
        got_name = rxlib.last_dispatch()
    finally:
        for cm in reversed(opts):
            cm.__exit__(None, None, None)
    assert got_name.split("|")[0] == c["expect"], (got_name, c["expect"])
    if c.get("runtime"):
        assert c["runtime"] in got_name.split("|")[1].split(","), got_name
    # fp8 pools: the rows are upcast exactly, so the result carries the one output rounding like a 16-bit pool's
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("dispatch", c["expect"]), ulps=1, absw=absw)


def _run_extend_unified(c, ops, rxlib):
    """The one-stage extend over the unified kv list (prefix slots + the new tokens' slots, all rows in the pool)."""
    dtype = DT[c["dt"]]
    hq, hkv, d, ps = c["hq"], c["hkv"], c["dk"], 16
    pre, ext = np.asarray(c["pre"], np.int64), np.asarray(c["ext"], np.int64)
    bs, T = len(pre), int(ext.sum())
    rng = np.random.default_rng(zlib.crc32(c["expect"].encode()))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    n_pages = int(sum(-(-int(p + e) // ps) for p, e in zip(pre, ext))) + 3
    page_ids = rng.permutation(np.arange(1, n_pages))
    kvi, kvp, pi = [], [0], 0
    for p_, e_ in zip(pre, ext):
        n = int(p_ + e_)
        npg = -(-n // ps)
        kvi.append((page_ids[pi: pi + npg, None] * ps + np.arange(ps)[None]).reshape(-1)[:n])
        pi += npg
        kvp.append(kvp[-1] + n)
    kvi = np.concatenate(kvi).astype(np.int64)
    kb = torch.randn(n_pages * ps, hkv, d, generator=g).to(dtype)
    vb = torch.randn(n_pages * ps, hkv, d, generator=g).to(dtype)
    q = torch.randn(T, hq, d, generator=g).to(dtype)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    sm = d ** -0.5
    want, absw = orc.extend_attention_unified(_bits(q), _bits(kb), _bits(vb), qo, np.asarray(kvp, np.int32), kvi, pre, sm_scale=sm,
                                              return_absw=True)
    kd, vd, page, lay, _ = _device_pool(ops, kb, vb, c["lin"], ps, hnd=(len(c["expect"]) % 2 == 0))
    o = torch.full((T, hq, d), float("nan"), dtype=dtype, device=DEV)
    idt = torch.int32 if c["idx"] == "int" else torch.int64
    opts = [rxlib.option(k, v) for k, v in c["opts"].items()]
    for cm in opts:
        cm.__enter__()
    try:
        ops.extend_attention_fwd_unified(q.to(DEV), o, kd, vd, 1.0, 1.0, torch.from_numpy(qo).to(DEV), torch.tensor(kvp, dtype=torch.int32, device=DEV),
                                         torch.from_numpy(kvi).to(DEV).to(idt), torch.from_numpy(pre.astype(np.int32)).to(DEV), int(ext.max()),
                                         sm_scale=sm, is_causal=True, page_size=page, kv_layout=lay)
        torch.cuda.synchronize()
        got_name = rxlib.last_dispatch()
    finally:
        for cm in reversed(opts):
            cm.__exit__(None, None, None)
    assert got_name.split("|")[0] == c["expect"], (got_name, c["expect"])
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("dispatch", c["expect"]), ulps=1, absw=absw)


def _run_decode(c, ops, rxlib):
    dtype = DT[c["dt"]]
    hq, hkv, dk, dv, ps = c["hq"], c["hkv"], c["dk"], c["dv"], 16
    lens = np.array([1, 31, 33, 257, 900, 64, 2100], dtype=np.int64)
    bs = len(lens)
    rng = np.random.default_rng(zlib.crc32((c["expect"] + c.get("runtime", "")).encode()))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    pages = [-(-int(n) // ps) for n in lens]
    n_pages = sum(pages) + 3
    page_ids = rng.permutation(np.arange(1, n_pages))
    r2t = np.zeros((bs + 1, int(lens.max()) + ps), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        r2t[i + 1, : int(n)] = (page_ids[pi: pi + pages[i], None] * ps + np.arange(ps)[None]).reshape(-1)[: int(n)]
        pi += pages[i]
    pool = n_pages * ps
    fp8 = c.get("fp8", False)
    pdt = torch.float8_e4m3fn if fp8 else dtype
    kb = torch.randn(pool, hkv, dk, generator=g).to(pdt)
    vb = kb[..., :dv] if c.get("mla") else torch.randn(pool, hkv, dv, generator=g).to(pdt)
    q = torch.randn(bs, hq, dk, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    sm = dk ** -0.5
    ip, ii = orc.build_kv_indices(r2t, rpi, lens)
    kn = vn = None
    if c.get("fuse"):  # the step's new K/V row, stored by the decode launch itself: the pool's row is poisoned first
        kn = torch.randn(bs, hkv, dk, generator=g).to(dtype)
        vn = torch.randn(bs, hkv, dv, generator=g).to(dtype)
        last = torch.from_numpy(r2t[rpi, lens - 1].astype(np.int64))
        kb_dev, vb_dev = kb.clone(), vb.clone()
        kb_dev[last], vb_dev[last] = 7.0, -7.0
        kb[last], vb[last] = kn, vn
    else:
        kb_dev, vb_dev = kb, vb
    if fp8:
        kbo = orc.fp8_e4m3fn_decode(kb.view(torch.uint8).numpy())
        vbo = kbo[..., :dv] if c.get("mla") else orc.fp8_e4m3fn_decode(vb.view(torch.uint8).numpy())
    else:
        kbo, vbo = _bits(kb), _bits(vb.contiguous())
    ks, vs = (0.8, 1.25) if (fp8 and not c.get("mla")) else (1.0, 1.0)
    aux = (1.5 * torch.randn(bs, hq, 70, generator=g)).to(dtype if len(c["expect"]) % 2 else torch.float32) if c.get("bias") else None
    okw = dict(k_scale=ks, v_scale=vs, score_bias=None if aux is None else aux.double().numpy())
    want, absw = orc.decode_attention(_bits(q), kbo, vbo, ip, ii, sm, return_absw=True, **okw)
    if c.get("mla"):
        if c["lin"]:
            kd = kb_dev.to(DEV)
            vd, page, lay = kd[..., :dv], 1, None
        else:
            kd = torch.zeros(n_pages, 1, ps + 2, dk, dtype=pdt)
            kd[:, :, :ps] = kb_dev.view(n_pages, ps, 1, dk).permute(0, 2, 1, 3)
            kd = kd.to(DEV)[:, :, :ps]
            vd, page = kd[..., :dv], ps
            lay = ops.kv_layout_hnd(kd, vd)
    else:
        kd, vd, page, lay, k_rows = _device_pool(ops, kb_dev, vb_dev, c["lin"], ps, hnd=(len(c["expect"]) % 2 == 0))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    S = 8
    lens_d = T(lens)
    splits = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits_balanced(splits, lens_d, hq, hkv, S, 64, 128)  # (a small budget: the long requests are cut)
    assert int(splits.max()) > 1
    al = torch.zeros(bs, hq, S, dv, dtype=torch.float32, device=DEV)
    ls = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o = torch.full((bs, hq, dv), float("nan"), dtype=dtype, device=DEV)
    mfma = c["expect"].startswith("decode_mfma_kernel") or c["expect"].startswith("decode_mfma_bias_kernel")
    order = torch.argsort(lens_d, descending=True).to(torch.int32) if mfma else None
    items = None
    if mfma and (c.get("occ3") or len(c["expect"]) % 3 == 0):
        items = ops.SplitItems(int(splits.clamp_min(1).sum()), DEV).build(splits, order, wgs_per_cu=3 if c.get("occ3") else 0)
    extra = dict(request_order=order, split_items=items)
    if c.get("fuse"):
        extra.update(k_new=kn.to(DEV), v_new=vn.to(DEV))
    if aux is not None:
        extra.update(score_mod=ops.relative_bias_score_mod, aux_tensors=[aux.to(DEV)])
    opts = [rxlib.option(k, v) for k, v in c.get("opts", {}).items()]
    for cm in opts:
        cm.__enter__()
    try:
        if c["mode"] == "indices":
            ops.decode_attention_fwd(q.to(DEV), kd, vd, o, T(ip), T(ii).to(c.get("idx_dtype", torch.int64)), al, ls, splits, S, sm,
                                     ks, vs, page_size=page, kv_layout=lay, **extra)
        else:
            ops.decode_attention_fwd_paged(q.to(DEV), kd, vd, o, T(r2t), T(rpi), lens_d, al, ls, splits, S, sm, ks, vs,
                                           page_size=page, kv_layout=lay, **extra)
        torch.cuda.synchronize()
        got_name = rxlib.last_dispatch()
    finally:
        for cm in reversed(opts):
            cm.__exit__(None, None, None)
    assert got_name.split("|")[0] == c["expect"], (got_name, c["expect"])
    if mfma:
        assert got_name.split("|")[1] == ("pairs" if items is not None else "slots") + "," + (
            "indices" if c["mode"] == "indices" else "req_to_token"), got_name
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("dispatch", c["expect"]), ulps=1, absw=absw)
    if c.get("fuse"):  # the launch wrote the new rows where the page table says
        assert torch.equal(k_rows()[last.to(DEV)].cpu().view(torch.int16), kn.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[_cid(c) for c in CASES])
def test_instance_is_dispatched_and_matches_the_oracle(case):
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    {"extend": _run_extend, "extend_unified": _run_extend_unified}.get(case["fam"], _run_decode)(case, ops, rxlib)
