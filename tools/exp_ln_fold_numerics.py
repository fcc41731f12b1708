"""Numerics of folding LayerNorm into the consuming GEMM ("Next" (1); CPU only):
    reference path   y = fp16(LN_fp32(x)) @ W^T        (fp32 accumulate, fp16 out)
    folded           y = rstd * (x @ (gamma*W)^T - mean * s) + c,   s[n] = sum_k (gamma*W)[n][k],  c[n] = sum_k beta_k W[n][k]
both against the exact fp64 result, for rows whose mean is up to 100 standard deviations away from zero.
Result (round 3): both paths sit at the output's own fp16 rounding (max 4.5e-4, rms 2.9e-4 relative) for every |mean| / sigma
tried — the subtraction of mean * s from the fp32 accumulator costs nothing measurable."""
import torch
torch.manual_seed(0)
def run(M,K,N,mu_scale,sigma=1.0):
    x=(torch.randn(M,K)*sigma+torch.randn(M,1)*mu_scale).half()
    gamma=(1+0.2*torch.randn(K)).float(); beta=(0.1*torch.randn(K)).float()
    W=(torch.randn(N,K)*K**-0.5).half(); b=torch.zeros(N)
    xd=x.double()
    mean=xd.mean(1,keepdim=True); var=xd.var(1,unbiased=False,keepdim=True); rstd=(var+1e-5).rsqrt()
    exact=((xd-mean)*rstd*gamma.double()+beta.double())@W.double().t()
    # reference path: LN (fp32) -> fp16 -> GEMM fp32 acc -> fp16
    xf=x.float(); m32=xf.mean(1,keepdim=True); v32=xf.var(1,unbiased=False,keepdim=True); r32=(v32+1e-5).rsqrt()
    ln16=((xf-m32)*r32*gamma+beta).half()
    ref=(ln16.float()@W.float().t()).half()
    # folded: raw x @ (gamma*W) fp16, epilogue
    Wp=(W.float()*gamma).half()
    s=Wp.float().sum(1); c=(W.float()*beta).sum(1)
    acc=x.float()@Wp.float().t()
    fold=(r32*(acc-m32*s)+c).half()
    def err(a): return ((a.double()-exact).abs().max()/exact.abs().max()).item(), ((a.double()-exact).pow(2).mean().sqrt()/exact.pow(2).mean().sqrt()).item()
    return err(ref), err(fold)
for mu in (0,1,5,20,100):
    r,f=run(4096,320,960,mu)
    print("mu/sigma=%5.0f  ref: max %.2e rms %.2e | folded: max %.2e rms %.2e"%(mu,r[0],r[1],f[0],f[1]))
for mu in (5,20):
    r,f=run(2048,1280,1280,mu)
    print("K=1280 mu/sigma=%3.0f  ref: max %.2e rms %.2e | folded: max %.2e rms %.2e"%(mu,r[0],r[1],f[0],f[1]))
