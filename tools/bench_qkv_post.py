import os, sys, torch, math
sys.path.insert(0, "/root/repo")
from unitex_amd.flux.transformer import FluxDiT, FluxShape
from unitex_amd.flux import ops
from unitex_amd._lib import QkvPostDesc, ptr
import ctypes as C
dev="cuda:0"; S=50688; H=24; D=3072
qkv=torch.randn(S,3*D,device=dev).to(torch.bfloat16); w=torch.ones(128,device=dev,dtype=torch.bfloat16)
cos=torch.rand(S,64,device=dev); sin=torch.rand(S,64,device=dev)
Qh=torch.zeros(H,S,128,device=dev,dtype=torch.bfloat16); Kh=torch.zeros_like(Qh); Vt=torch.zeros(H,128,S,device=dev,dtype=torch.bfloat16)
d=QkvPostDesc(); d.qkv,d.ld,d.q_col,d.k_col,d.v_col=ptr(qkv),qkv.stride(0),0,D,2*D
d.wq,d.wk,d.cosb,d.sinb=ptr(w),ptr(w),ptr(cos),ptr(sin); d.Qh,d.Kh,d.Vt=ptr(Qh),ptr(Kh),ptr(Vt)
d.hs_qk,d.hs_v,d.S_pad=Qh.stride(0),Vt.stride(0),S; d.n_tok,d.tok_off,d.H,d.eps,d.q_scale=S,0,H,1e-6,0.1275
ctx=ops.get_ctx(0)
def run(): ctx.check(ctx.lib.utx_qkv_post(ctx.handle,C.byref(d),ctx.stream()))
for _ in range(3): run()
torch.cuda.synchronize(); ts=[]
for _ in range(10):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort(); ms=ts[5]; gb=(S*3*D*2*2+S*64*8)/1e9
print("qkv_post S=%d: %.3f ms  %.0f GB/s" % (S, ms, gb/ms*1e3))
