"""C -> 1 brick forward at extra channel counts (32, 96, 512) and with / without the packed-weight workspace, against the gather kernel (float32 weights: the brick form rounds them to bf16, ~1e-3 relative)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pcrlv2_amd._lib import dtype_code, lib, stream_handle
L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
torch.manual_seed(0)
for (N,D,H,W,C) in ((2,4,8,8,32),(3,8,16,24,512),(2,12,8,16,96),(1,4,8,8,64)):
    M=N*D*H*W
    x=torch.randn(M,C,device=dev).to(dt); w=torch.randn(C,27,device=dev)*0.1; b=torch.randn(1,device=dev)
    outs=[]
    for impl,use_ws in ((0,True),(0,False),(1,False)):
        L.debug_set_conv_impl(impl)
        y=torch.zeros(M,device=dev); rows=L.call("pcrl_conv3d_to1_stats_rows",N,D,H,W,C,27,dtype_code(dt)); st=torch.zeros(rows*2,device=dev)
        nb=L.call("pcrl_conv3d_to1_fwd_ws_bytes",N,D,H,W,C,27); ws=torch.empty(nb,dtype=torch.uint8,device=dev)
        L.call("pcrl_conv3d_to1_fwd",x,w,b,y,st,ws if use_ws else None,nb if use_ws else 0,N,D,H,W,C,27,dtype_code(dt),stream_handle())
        torch.cuda.synchronize(); outs.append((y.clone(), st.view(-1,2).sum(0).clone()))
    L.debug_set_conv_impl(0)
    ref=outs[2][0]; wq=w.to(dt).float()
    e1=(outs[0][0]-ref).abs().max().item(); e2=(outs[1][0]-outs[0][0]).abs().max().item()
    print(f"N={N} {D}x{H}x{W} C={C}: brick(ws) vs gather max|d|={e1:.3e} (ref max {ref.abs().max():.2f}); brick ws vs no-ws {e2:.1e}; stats {outs[0][1].tolist()} vs {outs[2][1].tolist()}")
