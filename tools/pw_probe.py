import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pcrlv2_amd._lib import dtype_code, lib, stream_handle
L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
N,D,H,W,C=32,64,64,32,64
M=N*D*H*W
x=torch.randn(M,C,device=dev).to(dt); w=torch.randn(C,1,device=dev)*0.1; b=torch.zeros(1,device=dev)
y=torch.empty(M,device=dev)
def run(): L.call("pcrl_conv3d_to1_fwd",x,w,b,y,None,None,0,N,D,H,W,C,1,dtype_code(dt),stream_handle())
run(); torch.cuda.synchronize(); ts=[]
for _ in range(9):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort(); print(f"to1 pointwise fwd: {ts[4]*1e3:.1f} us  {M*C*2/1e9/ts[4]:.2f} TB/s  checksum {float(y.sum()):.3f}")
