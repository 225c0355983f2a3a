import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gava_clip_amd import hip
d=torch.device("cuda"); R,D=100864,768
x=torch.randn(R,D,device=d); g1,b1,g2,b2=(torch.randn(D,device=d) for _ in range(4))
o16=torch.empty(R,D,dtype=torch.float16,device=d); hi=torch.empty_like(o16); lo=torch.empty_like(o16); y32=torch.empty_like(x)
def t(fn,n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
P=hip.PREC_F16
print("plain LN -> h16            %.4f ms" % t(lambda: hip.layernorm(x,g1,b1,out16=o16,prec=P)))
print("prefused -> h16 + fp32     %.4f ms" % t(lambda: hip.layernorm(x,g1,b1,out16=o16,out32=y32,prec=P,gamma2=g2,beta2=b2)))
print("prefused -> h16 + hi + lo  %.4f ms" % t(lambda: hip.layernorm(x,g1,b1,out16=o16,prec=P,gamma2=g2,beta2=b2,out_hi=hi,out_lo=lo)))
