"""Split-schedule policies across batch shapes: slots grid at 2 WGs per CU vs the live-pairs grid (3 per CU)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops
dev="cuda"; HQ,HKV,D,PS=32,8,128,16
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
    byt=sum(lens)*HKV*D*2*2
    order=torch.argsort(lens_d,descending=True).to(torch.int32)
    def t(wg, items, mint, mixed=0):
        hc=ops.balanced_kv_splits_host(np.asarray(lens),HQ,HKV,64,wg,mint,mixed)
        S=int(hc.max())
        if S<=1: return None
        S8=(S+7)//8*8
        ns=torch.from_numpy(hc).to(dev)
        cnt=torch.zeros(bs*HQ,dtype=torch.int32,device=dev)
        si=ops.SplitItems(int(hc.sum()),dev).build(ns,order,wgs_per_cu=3 if mixed else 0) if items else None
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
        us=e0.elapsed_time(e1)/50*1e3
        return f"{us:5.0f}us S={S:2d} n={int(hc.sum()):3d}"
    blocks=bs*HKV
    mint=1024 if 2*blocks>=256 else 128
    print(f"{name:28s} {byt/1e6:6.0f}MB | slots wg512: {t(512,False,mint)} | items wg512: {t(512,True,mint)} | wg768: {t(768,True,mint)} | wg512/mixed768: {t(512,True,mint,768)}")
case([32768]+[1024]*63,"1x32k+63x1k")
case([8192]*4+[512]*124,"4x8k+124x512")
case(list(np.random.default_rng(1).integers(100,6000,size=96)),"ragged96")
case([16384]*2+[2048]*30,"2x16k+30x2k")
case([4096]*16,"16x4k"); case([8192]*8,"8x8k"); case([1024]*32,"32x1k"); case([32768],"1x32k"); case([16384]*4,"4x16k"); case([2048]*48,"48x2k")
case([65536]+[4096]*31,"1x64k+31x4k")
