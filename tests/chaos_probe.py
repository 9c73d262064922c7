"""How much do the bf16 gradients of the golden b=4 step move between equally valid kernel choices (summation orders)?
BatchNorm1d over four samples sits between the cosine losses and the decoder: see tests/test_model_gpu.py."""
import os, sys, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # lives under tests/: it uses the oracle helpers (test infrastructure)
sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/oracle")
import numpy as np, torch
import pcrlv2_oracle as O
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd._lib import lib
from test_model_gpu import build, forward_losses
fx = np.load(R + "/tests/golden/c_small_b4_32x32x16.npz")
b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
batches = [O.fill_batch(b, dhw, dtype=torch.float32, seed=7 + 100 * s) for s in range(int(fx["meta/nsteps"]))]
names = ["up_tr64.up_conv.weight", "up_tr64.bn.weight", "up_tr128.up_conv.weight", "down_tr64.ops.0.conv1.weight", "up_tr256.up_conv.weight"]
for dt in (torch.float32, torch.bfloat16):
    for impl in (0, 1, 2):
        lib().debug_set_conv_impl(impl)
        model = build(dt)
        r = forward_losses(model, batches[0], int(fx["meta/epoch"]), int(fx["meta/seed"]))
        r["loss"].backward()
        g = dict(model.named_parameters())
        print(dt, "impl", impl, "loss2 %.5f" % float(r["loss2"]), " ".join("%s %.3f" % (n.split(".")[0] + "." + n.split(".")[-2], float(g[n].grad.double().norm()) / float(fx[f"grad/{n}/l2"])) for n in names))
lib().debug_set_conv_impl(0)
