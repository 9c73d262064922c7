"""Is the 3D step host-bound?  Enqueue N steps without ever synchronising (config.MAX_STEPS_AHEAD lifted) and compare the host's enqueue time per
step with the GPU's time per step; then the default throttle.  Prints both, and how far the host was ahead when the last step was enqueued."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from pcrlv2_amd import config
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd import train_3d as T
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss
dev = torch.device("cuda")
torch.manual_seed(0); random.seed(0)
model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
batch = synthetic_batch(32, (64, 64, 32), 16, dev, 1)
crit, cos = MSELoss(), CosineSimilarityMean()
for _ in range(8):
    T.train_step(model, opt, batch, 0, crit, cos, guard=False)
torch.cuda.synchronize()
for lag in (100, config.MAX_STEPS_AHEAD, 1):
    config.MAX_STEPS_AHEAD = lag
    N = 20
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    hs = []
    for _ in range(N):
        a = time.perf_counter()
        T.train_step(model, opt, batch, 0, crit, cos, guard=False)
        hs.append(time.perf_counter() - a)
    t_host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    hs.sort()
    print(f"MAX_STEPS_AHEAD={lag}: host enqueue {1e3 * t_host / N:.2f} ms/step (median call {1e3 * hs[N // 2]:.2f}, min {1e3 * hs[0]:.2f}), GPU {e0.elapsed_time(e1) / N:.2f} ms/step, "
          f"wall {1e3 * t_all / N:.2f} ms/step; host finished enqueueing {1e3 * (t_all - t_host):.1f} ms before the GPU finished")
