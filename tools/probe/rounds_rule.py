"""Two-per-CU schedules for mixed batches: first pass (t* = even share of 2 x CUs) vs the 'rounds' rule (long pieces =
R x the mean unsplit length, R = rounds of workgroups the batch needs on 2 x CUs slots)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops
dev="cuda"; HQ,HKV,D,PS=32,8,128,16
Z=512
def rounds_rule(lens, first):
    lens=np.asarray(lens); split=first>1
    if not split.any() or split.all(): return first
    a=float(lens[~split].mean()); wu=int((~split).sum())*HKV
    for R in (1,2,3,4):
        p=R*a
        n=np.where(split, np.ceil(lens/p), 1).astype(np.int64)
        tot=wu+int(n[split].sum())*HKV
        if -(-tot//Z) <= R: break
    return np.minimum(n,64).astype(np.int32)
def case(lens, name):
    bs=len(lens); ctx=int(max(lens))
    pages=[(n+PS-1)//PS for n in lens]
    rng=np.random.default_rng(0)
    perm=rng.permutation(np.arange(1,sum(pages)+1))
    r2t=np.zeros((bs+1,ctx+PS),dtype=np.int32); pi=0
    for i,n in enumerate(lens):
        sl=(perm[pi:pi+pages[i],None]*PS+np.arange(PS)[None]).reshape(-1)[:n]; pi+=pages[i]; r2t[i+1,:n]=sl
    pool=(sum(pages)+1)
    kb=torch.randn(pool,HKV,PS,D,device=dev).to(torch.bfloat16); vb=torch.randn_like(kb)
    lay=ops.kv_layout_hnd(kb,vb)
    q=torch.randn(bs,HQ,D,device=dev).to(torch.bfloat16); o=torch.empty_like(q)
    r2td=torch.from_numpy(r2t).to(dev); rpi=torch.arange(1,bs+1,device=dev); lens_d=torch.tensor(lens,dtype=torch.int64,device=dev)
    order=torch.argsort(lens_d,descending=True).to(torch.int32)
    def t(hc, occ3=False):
        S=int(hc.max())
        if S<=1: return None
        S8=(S+7)//8*8
        ns=torch.from_numpy(hc).to(dev)
        cnt=torch.zeros(bs*HQ,dtype=torch.int32,device=dev)
        si=ops.SplitItems(int(hc.sum()),dev).build(ns,order,wgs_per_cu=3 if occ3 else 0)
        al=torch.empty(bs,HQ,S8,D,dtype=torch.float32,device=dev); lse=torch.empty(bs,HQ,S8,device=dev)
        def f():
            ops.decode_attention_fwd_paged(q,kb,vb,o,r2td,rpi,lens_d,al,lse,ns,S8,D**-0.5,page_size=PS,kv_layout=lay,merge_counters=cnt,request_order=order,split_items=si)
        for _ in range(3): f()
        torch.cuda.synchronize()
        st=torch.cuda.Stream()
        with torch.cuda.stream(st):
            f(); gr=torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10): f()
            gr.replay(); torch.cuda.synchronize()
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): gr.replay()
            e1.record(); torch.cuda.synchronize()
        return f"{e0.elapsed_time(e1)/50*1e3:5.0f}us S={S:2d} n={int(hc.sum()):3d}"
    mint=1024 if 2*bs*HKV>=256 else 128
    first=ops.balanced_kv_splits_host(np.asarray(lens),HQ,HKV,64,512,mint)
    rr=ops.balanced_kv_splits_host(np.asarray(lens),HQ,HKV,64,512,mint,-1)
    dv=torch.zeros(bs,dtype=torch.int32,device=dev); ops.get_num_kv_splits_balanced(dv,lens_d,HQ,HKV,64,512,mint,-1); assert dv.cpu().numpy().tolist()==rr.tolist(), (dv.cpu().numpy()[:4], rr[:4])
    mixed=ops.balanced_kv_splits_host(np.asarray(lens),HQ,HKV,64,512,mint,768)
    print(f"{name:24s} | 2/CU first pass: {t(first)} | 2/CU rounds rule: {t(rr)} | 3/CU mixed: {t(mixed,True)}")
case([32768]+[1024]*63,"1x32k+63x1k")
case([8192]*4+[512]*124,"4x8k+124x512")
case([16384]*2+[2048]*30,"2x16k+30x2k")
case([65536]+[4096]*31,"1x64k+31x4k")
case([32768]+list(np.random.default_rng(2).integers(300,2000,size=63)),"1x32k+63 ragged")
case([16384,12000]+list(np.random.default_rng(3).integers(500,3000,size=94)),"2 long+94 ragged")
case([40000]+[2048]*127,"1x40k+127x2k")
