#!/usr/bin/env python3
"""numpy emulation of the int8 digit-plane product of csrc/gemm_i8.hip (A = W Kzx with W = chol(Kzz)^-1 in float64): for
several (planes of W, planes of Kzx, highest level kept) it prints the number of int8 plane products and the error against
the float64 product, next to the float32 product and the float64-accumulating product on a float32 Kzx.  CPU only; this
is how 5 x 4 planes / 14 products (and 5 x 5 / 19 for layers that feed the next) were chosen.

    python tools/probes/ozaki_emulation.py
"""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench


def rbf_ard(a, b, ls, os_):
    d = (a[:, None, :] - b[None, :, :]) / ls
    return os_ * torch.exp(-0.5 * (d * d).sum(-1))


torch.manual_seed(173)
M=1024
x_all,y_all=bench.synthetic_grid()
g=torch.Generator().manual_seed(5)
rows=torch.randperm(100000,generator=g)[:1024]
x=x_all[rows].double()
# hidden-layer-like GP: Z ~ randn (bench init), ls=softplus(0), os=softplus(0)
Z=torch.randn(M,3,generator=g,dtype=torch.float64)
ls=torch.full((1,3),math.log(2.0),dtype=torch.float64); os_=math.log(2.0)
Kzz=rbf_ard(Z,Z,ls,os_)+1e-4*torch.eye(M,dtype=torch.float64)
L=torch.linalg.cholesky(Kzz); W=torch.linalg.inv(L)
Kzx=rbf_ard(Z,x,ls,os_)                  # (M,n) f64
A_exact=(W@Kzx)
print('kappa',float(torch.linalg.cond(Kzz)),'max|W|',float(W.abs().max()),'max|A|',float(A_exact.abs().max()),
      'max sum|W||K|',float((W.abs()@Kzx.abs()).max()))
A32=(W.float()@Kzx.float()).double()
print('f32 product err rel max|A|:',float((A32-A_exact).abs().max()/A_exact.abs().max()))
Kf=Kzx.float().double()    # f32-rounded Kzx, f64 accumulate (the current f64acc path)
print('f64acc on f32 Kzx err:',float((W@Kf-A_exact).abs().max()/A_exact.abs().max()))
def slices(X,scale,s):
    t=(X/scale)*64.0
    out=[]
    for i in range(s):
        d=np.rint(t); out.append(d); t=(t-d)*128.0   # digits kept as float64: products < 2^53 are exact, and BLAS runs them
    return out
Wn=W.numpy(); Kn=Kzx.numpy()
rs=2.0**np.ceil(np.log2(np.abs(Wn).max(axis=1,keepdims=True)))
cs=2.0**math.ceil(math.log2(os_))      # K <= os
for sW,sK,cut in ((5,5,5),(5,5,4),(6,6,5),(6,5,5),(5,4,4),(6,4,5),(4,4,3)):
    dW=slices(Wn,rs,sW); dK=slices(Kn,cs,sK)
    acc=np.zeros((M,Kn.shape[1]))
    npairs=0
    for l in range(cut+1):
        lev=np.zeros((M,Kn.shape[1]))
        for a in range(sW):
            b=l-a
            if 0<=b<sK:
                lev+=dW[a]@dK[b]; npairs+=1
        assert np.abs(lev).max()<2**31
        acc+=lev*128.0**(-l)
    A=acc*rs*cs/4096.0
    err=np.abs(A-A_exact.numpy()).max()/np.abs(A_exact.numpy()).max()
    print(f'slices W {sW} K {sK} levels<= {cut}: pairs {npairs} err rel max|A| {err:.3g}')
