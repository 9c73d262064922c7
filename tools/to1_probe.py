"""Probe of the 1-channel layers: C -> 1 brick forward (deep-supervision heads) and 1 -> Co first layer, HIP-event timing at the C2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcrlv2_amd._lib import dtype_code, lib, stream_handle
L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
def timed(fn, n=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for (N,D,H,W,C) in ((32,64,64,32,64),(32,32,32,16,128),(32,16,16,8,256),(192,16,16,16,64)):
    M=N*D*H*W
    x=torch.randn(M,C,device=dev).to(dt); w=torch.randn(C,27,device=dev)*0.1; b=torch.zeros(1,device=dev)
    y=torch.empty(M,device=dev); rows=L.call("pcrl_conv3d_to1_stats_rows",N,D,H,W,C,27,dtype_code(dt)); st=torch.empty(rows*2,device=dev)
    nb=L.call("pcrl_conv3d_to1_fwd_ws_bytes",N,D,H,W,C,27); ws=torch.empty(nb,dtype=torch.uint8,device=dev)
    t=timed(lambda: L.call("pcrl_conv3d_to1_fwd",x,w,b,y,st,ws,nb,N,D,H,W,C,27,dtype_code(dt),stream_handle()))
    print(f"to1 fwd N={N} {D}x{H}x{W} C={C}: {t*1e3:7.1f} us  ({M*C*2/1e9/t:.2f} TB/s of x)")
for (N,D,H,W,Co) in ((32,64,64,32,32),(192,16,16,16,32)):
    M=N*D*H*W
    x=torch.randn(N,1,D,H,W,device=dev); w=torch.randn(Co,1,3,3,3,device=dev)*0.2; b=torch.zeros(Co,device=dev)
    y=torch.empty(M,Co,device=dev,dtype=dt); rows=L.call("pcrl_conv3d_k3_c1_stats_rows",N,D,H,W,Co,dtype_code(dt)); st=torch.empty(rows*Co*2,device=dev)
    t=timed(lambda: L.call("pcrl_conv3d_k3_c1_fwd",x,w,b,y,st,N,D,H,W,Co,dtype_code(dt),stream_handle()))
    print(f"c1 fwd N={N} {D}x{H}x{W} Co={Co}: {t*1e3:7.1f} us  ({M*Co*2/1e9/t:.2f} TB/s of y)")
