"""Dev experiment: decode time of heterogeneous batches (one long request among short ones, two length classes, a ragged
batch) under a single pass, the reference's K3 split formula and a length-balanced split count.  python tools/hetero_decode.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops
dev="cuda"; HQ,HKV,D,PS=32,8,128,16
def case(lens):
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
    def t(ns, S, order=None, items=False):
        cnt=torch.zeros(bs*HQ,dtype=torch.int32,device=dev)
        si=ops.SplitItems(int(ns.clamp_min(1).sum()),dev).build(ns,order,wgs_per_cu=3 if items==3 else 0) if items else None
        al=torch.empty(bs,HQ,S,D,dtype=torch.float32,device=dev); lse=torch.empty(bs,HQ,S,device=dev)
        def f():
            if S==1: ops.decode_attention_fwd_paged(q,kb,vb,o,r2td,rpi,lens_d,None,None,None,1,D**-0.5,page_size=PS,kv_layout=lay,request_order=order)
            else: ops.decode_attention_fwd_paged(q,kb,vb,o,r2td,rpi,lens_d,al,lse,ns,S,D**-0.5,page_size=PS,kv_layout=lay,merge_counters=cnt,request_order=order,split_items=si)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/10*1e3
        return f"{us:.0f} us {byt/us/1e6:.2f} TB/s"
    order=torch.argsort(lens_d,descending=True).to(torch.int32)
    print("lens", f"bs={bs} max={max(lens)} sum={sum(lens)}")
    print("  single pass           ", t(None,1), "| ordered", t(None,1,order))
    for S in (8,16,32):
        ref=torch.zeros(bs,dtype=torch.int32,device=dev); ops.get_num_kv_splits(ref,lens_d.int(),HQ,HKV,S,256)
        print(f"  K3 formula, max {S:2d}     ", t(ref,S), ref.tolist()[:3], int(ref.max()))
        tot=sum(lens)*HKV; tstar=max(512, tot/512.0)
        bal=torch.tensor([min(S,max(1,int(np.ceil(n/tstar)))) for n in lens],dtype=torch.int32,device=dev)
        print(f"  balanced, max {S:2d}       ", t(bal,S,order), "| live pairs only:", t(bal,S,order,True), bal.tolist()[:3], int(bal.max()))
case([32768]+[1024]*63)
case([8192]*4+[512]*124)
case(list(np.random.default_rng(1).integers(100,6000,size=96)))
