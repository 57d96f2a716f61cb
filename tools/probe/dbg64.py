import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops
dev="cuda"
def run(prefix, extend, hq, hkv, d=64):
    g=torch.Generator().manual_seed(1)
    P=sum(prefix); T=sum(extend)
    kb=torch.randn(P+8,hkv,d,generator=g).to(torch.bfloat16); vb=torch.randn(P+8,hkv,d,generator=g).to(torch.bfloat16)
    q=torch.randn(T,hq,d,generator=g).to(torch.bfloat16); ke=torch.randn(T,hkv,d,generator=g).to(torch.bfloat16); ve=torch.randn(T,hkv,d,generator=g).to(torch.bfloat16)
    kvp=np.concatenate([[0],np.cumsum(prefix)]).astype(np.int32); qo=np.concatenate([[0],np.cumsum(extend)]).astype(np.int64)
    kvi=torch.arange(1,P+1,dtype=torch.int64)
    outs=[]
    for env in (None,"1"):
        if env: os.environ["RX_EXT_D256_AT64"]=env
        else: os.environ.pop("RX_EXT_D256_AT64",None)
        o=torch.full((T,hq,d),float("nan"),dtype=torch.bfloat16,device=dev)
        ops.extend_attention_fwd(q.to(dev),ke.to(dev),ve.to(dev),o,kb.to(dev),vb.to(dev),torch.from_numpy(qo).to(dev),torch.from_numpy(kvp).to(dev),kvi.to(dev),None,True,None,max(extend),1.0,1.0,sm_scale=d**-0.5,page_size=1)
        torch.cuda.synchronize(); outs.append(o.float().cpu().numpy())
    diff=np.abs(outs[0]-outs[1])
    bad=np.argwhere(diff>0.05)
    print(prefix,extend,hq,hkv,"max diff",diff.max(),"nbad",len(bad))
    if len(bad):
        print(" bad tokens:",sorted(set(bad[:,0].tolist()))[:20],"heads",sorted(set(bad[:,1].tolist())),"cols",sorted(set(bad[:,2].tolist()))[:70])
# note: the env switch is read once per process (static) -> run each config in its own process via argv
cfg=int(sys.argv[1])
cases=[([0],[256],1,1),([0],[256],4,4),([300],[200],4,4),([0],[130,300],4,4),([64,100],[129,256],4,2)]
run(*cases[cfg])
