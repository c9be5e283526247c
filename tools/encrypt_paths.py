import sys, os, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import torch
from conftest import load_fixture
import bgn_amd, bgn_amd.synthetic as syn
dev=torch.device("cuda",0)
print("key,count,path,ms,kernel")
for key in ("k512","k1024"):
    fx=load_fixture(key)
    pk=bgn_amd.PublicKey(int(fx["p"],16),int(fx["n"],16),fx["l"],bytes.fromhex(fx["P"]),bytes.fromhex(fx["Q"]),fx["msg_space"],True,fx["poly_base"])
    eng=pk.engine; EB=eng.elem_bytes
    NMAX=1<<16
    g=torch.Generator().manual_seed(3)
    xs=torch.randint(0,256,(NMAX,3),dtype=torch.uint8,generator=g).to(dev)
    r_len,top_mask=syn._r_shape(int(fx["n"],16))
    rs=torch.randint(0,256,(NMAX,r_len),dtype=torch.uint8,generator=g); rs[:,0]&=top_mask; rs=rs.to(dev)
    out=torch.empty(NMAX*EB,dtype=torch.uint8,device=dev)
    ref=None
    for n in (1,64,1024,4096,16384,65536):
        res={}
        for path,val in (("chains",0),("lane groups",1<<20)):
            eng.set_option("quad_max_enc",val)
            best=1e9
            for _ in range(4):
                torch.cuda.synchronize(); t=time.perf_counter()
                eng.encrypt_dev(xs[:n],3,rs[:n],r_len,out[:n*EB],n)
                torch.cuda.synchronize(); best=min(best,time.perf_counter()-t)
            res[path]=out[:n*EB].clone()
            print("%s,%d,%s,%.3f,%s"%(key,n,path,best*1e3,eng.last_kernel_name()),flush=True)
        assert torch.equal(res["chains"],res["lane groups"])
