"""The divergence guard of the 3D step (reference train_3d.py:140-142: `if loss > 1000 and epoch > 10: continue` before zero_grad).

What the reference's `continue` leaves behind, and what is asserted here for both forms of the guard (decided on the device -- the default,
no forward -> backward host synchronisation -- and `guard="sync"`, the reference's host-side decision):
  * parameters and momentum buffers bit-unchanged;
  * BatchNorm running statistics moved by the step's three forwards, `num_batches_tracked` advanced;
  * the next step proceeds normally.
Plus: the guard is inactive up to epoch 10, a step that does not diverge is bit-identical to the unguarded step, and under data parallelism
the decision is collective (only rank 1 diverges: both ranks skip, nobody waits in an all-reduce that never comes)."""
import os
import random
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import pcrlv2_oracle as O  # noqa: E402
from pcrlv2_amd.models import PCRLv23d  # noqa: E402
from pcrlv2_amd.optim import FusedSGD  # noqa: E402
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step  # noqa: E402

DEV = "cuda"


def _setup(dt=torch.float32):
    random.seed(1)
    model = PCRLv23d().to(DEV)
    model.load_state_dict(O.fill_state(torch.float32))
    model.train().set_compute_dtype(dt)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    good = O.fill_batch(4, (16, 16, 16), dtype=torch.float32, seed=21)
    x1, x2, gt, gt2, loc = O.fill_batch(4, (16, 16, 16), dtype=torch.float32, seed=22)
    bad = (x1, x2, gt * 1e3, gt2, loc)          # sigmoid output vs a target ~1e3: MSE ~ 3e5 > 1000
    return model, opt, good, bad


def _buffers(model):
    return {k: v.clone() for k, v in model.state_dict().items() if O.is_buffer(k)}


@pytest.mark.parametrize("mode", [True, "sync"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_diverged_step_leaves_parameters_and_momentum_untouched(mode, dt):
    model, opt, good, bad = _setup(dt)
    crit, cosine = MSELoss(), CosineSimilarityMean()
    out = train_step(model, opt, good, 11, crit, cosine, guard=mode)          # momentum buffers exist from here on
    assert out is not None and (out.skipped is None or float(out.skipped) == 0.0)
    p0, m0, b0 = opt.flat_p.clone(), opt.flat_buf.clone(), _buffers(model)
    grads0 = [None if p.grad is None else p.grad.clone() for p in opt._plist]

    out = train_step(model, opt, bad, 11, crit, cosine, guard=mode)
    torch.cuda.synchronize()
    if mode == "sync":
        assert out is None                                                    # the reference's `continue`
        for g0, p in zip(grads0, opt._plist):                                 # ... which comes before zero_grad: gradients untouched too
            assert (g0 is None) == (p.grad is None) and (g0 is None or torch.equal(g0, p.grad))
    else:
        assert float(out[0]) > 1000 and float(out.skipped) == 1.0
    assert torch.equal(opt.flat_p, p0), "a skipped step changed parameters"
    assert torch.equal(opt.flat_buf, m0), "a skipped step changed momentum buffers"
    b1 = _buffers(model)
    moved = [k for k in b0 if k.endswith("running_mean") and not torch.equal(b0[k], b1[k])]
    assert len(moved) == len([k for k in b0 if k.endswith("running_mean")]), "running statistics must move in the three forwards of a skipped step"
    for k in b0:
        if k.endswith("num_batches_tracked"):
            assert int(b1[k]) == int(b0[k]) + 3, k

    out = train_step(model, opt, good, 11, crit, cosine, guard=mode)          # the next step proceeds
    torch.cuda.synchronize()
    assert out is not None and (out.skipped is None or float(out.skipped) == 0.0)
    assert not torch.equal(opt.flat_p, p0) and not torch.equal(opt.flat_buf, m0)
    assert all(torch.isfinite(l) for l in out)


def test_guard_is_inactive_up_to_epoch_10_and_free_when_not_diverged():
    crit, cosine = MSELoss(), CosineSimilarityMean()
    # epoch 10, loss > 1000: the reference updates (its condition needs epoch > 10)
    model, opt, good, bad = _setup()
    p0 = opt.flat_p.clone()
    out = train_step(model, opt, bad, 10, crit, cosine)
    assert out is not None and out.skipped is None and float(out[0]) > 1000
    assert not torch.equal(opt.flat_p, p0)
    # epoch 11, no divergence: the guarded step is the unguarded step, bit for bit
    finals = []
    for guard in (True, False, "sync"):
        model, opt, good, _ = _setup()
        random.seed(5)
        for _ in range(2):
            train_step(model, opt, good, 11, crit, cosine, guard=guard)
        finals.append((opt.flat_p.clone(), opt.flat_buf.clone()))
    for p, m in finals[1:]:
        assert torch.equal(p, finals[0][0]) and torch.equal(m, finals[0][1])


def test_first_update_after_a_skipped_first_step_is_a_plain_first_update():
    """Resume at epoch > 10 without optimizer state and diverge at once: the host marks the momentum buffers initialised although the device
    skipped; the buffers are zeros, so the next update (momentum * 0 + g) is the first update."""
    crit, cosine = MSELoss(), CosineSimilarityMean()
    model, opt, good, bad = _setup()
    train_step(model, opt, bad, 11, crit, cosine)                 # skipped on the device; host state says "initialised"
    random.seed(9)
    train_step(model, opt, good, 11, crit, cosine)
    want_p, want_m = opt.flat_p.clone(), opt.flat_buf.clone()
    model2, opt2, good2, _ = _setup()
    # reference run: the same three forwards of the bad batch (they move the running statistics), no update, then the good step
    from pcrlv2_amd.train_3d import begin_step, step_losses
    begin_step()
    step_losses(model2, bad, 11, crit, cosine)
    random.seed(9)
    train_step(model2, opt2, good2, 11, crit, cosine, guard=False)
    assert torch.equal(opt2.flat_p, want_p) and torch.equal(opt2.flat_buf, want_m)


GUARD2_WORKER = r'''
import os, sys, random, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import pcrlv2_oracle as O
from pcrlv2_amd import ddp
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step
rank, world, _ = ddp.init_process_group_from_env("gloo")     # two processes, ONE GPU
torch.cuda.set_device(0)
mode = True if sys.argv[2] == "device" else "sync"
random.seed(3)
model = PCRLv23d().cuda()
model.load_state_dict(O.fill_state(torch.float32))
model.train()
opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
dp = ddp.DataParallel(model, opt, bucket_mb=8.0)
crit, cosine = MSELoss(), CosineSimilarityMean()
good = O.fill_batch(2, (16, 16, 16), dtype=torch.float32, seed=40 + rank)
x1, x2, gt, gt2, loc = O.fill_batch(2, (16, 16, 16), dtype=torch.float32, seed=50 + rank)
bad = (x1, x2, gt * (1e3 if rank == 1 else 1.0), gt2, loc)      # ONLY rank 1 diverges
train_step(model, opt, good, 11, crit, cosine, guard=mode)
p0, m0 = opt.flat_p.clone(), opt.flat_buf.clone()
out = train_step(model, opt, bad, 11, crit, cosine, guard=mode)
torch.cuda.synchronize()
if mode == "sync":
    assert out is None, "rank %d did not skip" % rank
else:
    assert float(out.skipped) == 1.0, "rank %d did not skip" % rank
    assert (float(out[0]) > 1000) == (rank == 1)
assert torch.equal(opt.flat_p, p0) and torch.equal(opt.flat_buf, m0)
out = train_step(model, opt, good, 11, crit, cosine, guard=mode)     # nobody is stuck in a collective: the next step runs on both ranks
torch.cuda.synchronize()
assert out is not None and not torch.equal(opt.flat_p, p0)
mine = opt.flat_p.double().sum().reshape(1).cpu()
both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(both, mine)
assert both[0].item() == both[1].item(), both
dist.barrier()
print("OK", rank, flush=True)
dist.destroy_process_group()      # tear the group down before the interpreter exits (gloo threads alive at exit abort the process now and then)
'''


@pytest.mark.parametrize("mode", ["device", "sync"])
def test_guard_is_collective_two_ranks_one_gpu_gloo(tmp_path, mode):
    script = tmp_path / "guard2.py"
    script.write_text(GUARD2_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = "29771" if mode == "device" else "29773"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root, mode], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert all("OK" in o for o in outs)


# ---- every N-rank entry point exits clean (reference: train_3d.py:54's nn.DataParallel needs no teardown, one process per GPU does) ----
_ABORT_MARKS = ("terminate called", "destroy_process_group() was not called", "ChildFailedError", "Aborted", "SIGABRT")


def test_bench_one_rank_rccl_group_is_destroyed_before_exit():
    """`PCRL_FORCE_DDP=1 python bench.py`: the data-parallel wrapper on a ONE-rank RCCL group (real collectives on the communication stream).  The
    process must end rc = 0 with its JSON line and WITHOUT ProcessGroupNCCL's "destroy_process_group() was not called before program exit" warning
    -- bench.py leaves through ddp.shutdown() on every rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCRL_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--b", "4", "--dhw", "32,32,16", "--no-cpu-baseline",
                        "--no-secondary", "--no-alone"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert not any(m in r.stderr for m in _ABORT_MARKS), r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["distributed"]["forced_one_rank_probe"] is True and d["value"] > 0


@pytest.mark.parametrize("d", [3, 2])
def test_main_py_two_ranks_one_gpu_exits_clean(tmp_path, d):
    """`main.py --data synthetic` as two ranks (gloo, both on cuda:0 -- PCRL_DIST_BACKEND): real model, real steps, the product's own
    train_pcrlv2_3d / train_pcrlv2 creates the process group and destroys it in its `finally`.  Both ranks rc = 0, no abort message."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29781 + d), WORLD_SIZE="2", PCRL_DIST_BACKEND="gloo", PCRL_BIND_CPUS="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "main.py"), "--data", "synthetic", "--d", str(d), "--b", "4", "--epochs", "1", "--steps_per_epoch", "2",
           "--output", str(tmp_path), "--gpus", "0", "--amp"] + (["--size2d", "64"] if d == 2 else [])
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert not any(m in o for o in outs for m in _ABORT_MARKS), "\n".join(o[-3000:] for o in outs)
    assert "==> Saving..." in outs[0] and "total time" in outs[0]
