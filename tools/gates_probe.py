#!/usr/bin/env python3
"""Numbers behind two test gates (run on the GPU box):
  (1) float32 engine vs the float64 reference golden (c_small_b4_32x32x16, c_b16_32x32x16): per-tensor gradient rel-L2, sorted -- which
      tensors sit above SURVEY App. C's 5e-3 and by how much;
  (2) the census of bf16-typed C-ABI calls of one bf16 training step at the fixture size (entry point -> calls): every bf16 store of the
      engine happens inside one of them; tests/golden/bf16_call_census.json freezes it (tests/test_model_gpu.py::test_bf16_rounding_points_census).
      --write-census rewrites that file."""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import pcrlv2_oracle as O
from make_golden import sample_idx
from pcrlv2_amd import _lib
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step


def census(dtype=torch.bfloat16, b=4, dhw=(32, 32, 16)):
    L = _lib.lib()

    class C:
        watch = set(L.protos)
        calls = {}

        def add(self, name, args):
            a = {an: v for (_, an), v in zip(L.protos[name][1], args)}
            if a.get("dtype") == _lib.PCRL_BF16:
                self.calls[name] = self.calls.get(name, 0) + 1
    model = PCRLv23d().cuda()
    model.load_state_dict(O.fill_state(torch.float32))
    model.train().set_compute_dtype(dtype)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=7)
    random.seed(0)
    train_step(model, opt, batch, 3, MSELoss(), CosineSimilarityMean())      # first step: packs etc.
    torch.cuda.synchronize()
    c = C()
    L.counter = c
    random.seed(0)
    train_step(model, opt, batch, 3, MSELoss(), CosineSimilarityMean())
    torch.cuda.synchronize()
    L.counter = None
    return dict(sorted(c.calls.items()))


def grads_report(tag):
    fx = np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz"))
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    model = PCRLv23d().cuda()
    model.load_state_dict(O.fill_state(torch.float32))
    model.train().set_compute_dtype(torch.float32)
    opt = FusedSGD(model.parameters(), lr=float(fx["meta/lr"]), momentum=0.9, weight_decay=1e-4)
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=7)
    random.seed(int(fx["meta/seed"]))
    train_step(model, opt, batch, int(fx["meta/epoch"]), MSELoss(), CosineSimilarityMean())
    torch.cuda.synchronize()
    rows = []
    for name, p in model.named_parameters():
        if f"grad/{name}/none" in fx.files or p.grad is None:
            continue
        ref_s, l2 = fx[f"grad/{name}/samples"], float(fx[f"grad/{name}/l2"])
        if l2 < 1e-8:
            continue
        f = p.grad.detach().double().cpu().reshape(-1).numpy()
        got_s = f[sample_idx(f.size, 64, 3)]
        rows.append((float(np.linalg.norm(got_s - ref_s) / max(np.linalg.norm(ref_s), 1e-30)), abs(float(np.linalg.norm(f)) - l2) / l2, name, p.numel()))
    rows.sort(reverse=True)
    print(f"[{tag}] float32 engine vs float64 golden: {len(rows)} tensors; above 5e-3: {sum(1 for r in rows if r[0] > 5e-3)}")
    for r in rows[:25]:
        print("   rel-L2(samples) %.2e  |norm| %.2e  %-50s %d" % r)


if __name__ == "__main__":
    for tag in ("c_small_b4_32x32x16", "c_b16_32x32x16"):
        grads_report(tag)
    cz = census()
    print(json.dumps(cz, indent=1))
    if "--write-census" in sys.argv:
        path = os.path.join(ROOT, "gpurun_out", "bf16_call_census.json")
        json.dump({"fixture": "b=4, 32x32x16 + 6 x 16^3, bf16, one steady-state step (random.seed(0), epoch 3)", "calls": cz}, open(path, "w"), indent=1)
        print("wrote", path)
