"""Trilinear upsample backward at the two shapes of a C2 step (x4 of the 16x16x8 map, x2 of the 32x32x16 map) (the per-plane LDS form; the gather form it replaced was removed with its switch in round 6)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pcrlv2_amd._lib import lib, stream_handle
L, dev = lib(), torch.device("cuda")
def timed(fn, n=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for (N,D,H,W,s) in ((32,16,16,8,4),(32,32,32,16,2)):
    dy=torch.randn(N*D*s*H*s*W*s,device=dev); dx=torch.empty(N*D*H*W,device=dev)
    t=timed(lambda: L.call("pcrl_upsample_trilinear_bwd",dy,dx,N,D,H,W,s,stream_handle()))
    print(f"tri bwd x{s} {D}x{H}x{W}: {t*1e3:.1f} us")
