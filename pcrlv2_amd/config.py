"""Global knobs of the MI355X engine.

Round 6: switches whose verdict is final (every fusion / stream placement below that is bit-identical or measured better on every box) are plain
module constants -- the tests flip the attribute for their bit-identity checks; only what an operator of the engine may want to set from outside
(dtype, stream layout for profiling, run-ahead, allocator provisioning, the reference's per-epoch empty_cache) still reads the environment."""
import os

import torch

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
_default = _DTYPES[os.environ.get("PCRL_DTYPE", "fp32").lower()]


def default_compute_dtype() -> torch.dtype:
    """Activation / MFMA-operand dtype for newly built models: float32 (exact-parity mode, the reference's
    default precision) unless PCRL_DTYPE=bf16 or `set_default_compute_dtype` says otherwise.  The training
    entry point maps the reference's `--amp` switch (apex O1 fp16 there) to bfloat16."""
    return _default


def set_default_compute_dtype(dt):
    global _default
    _default = _DTYPES[dt.lower()] if isinstance(dt, str) else dt
    if _default not in (torch.float32, torch.bfloat16):
        raise ValueError(f"compute dtype must be float32 or bfloat16, got {dt}")


# Parameter gradients are delivered to `.grad` by the engine (pcrlv2_amd.functions) instead of autograd's AccumulateGrad nodes.
# PCRL_AUTOGRAD_PARAM_GRADS=1 hands them back to autograd (needed for torch.autograd.grad(loss, params), which never accumulates).
DIRECT_PARAM_GRADS = os.environ.get("PCRL_AUTOGRAD_PARAM_GRADS", "0") != "1"

# Weight-gradient kernels on a side stream, concurrently with the data-gradient / BatchNorm-backward chain of the layers below
# (MFMA-bound next to HBM-bound work); joined before the parameter gradients are summed (ops.side_wgrad).
#   PCRL_WGRAD_STREAM unset or 1: on (2D: +3.6 % at C5; 3D: +4 % at C2, 38.45 -> 36.95 ms same box);  =0: off everywhere.
#   With a neighbour on the chip a kernel's per-launch time is its time UNDER that contention (brick16 0.333 -> 0.346 ms): bench.py's
#   `roofline` reports the timed region as it ran and adds `roofline.alone` from a few extra steps with the side stream off.
_ws = os.environ.get("PCRL_WGRAD_STREAM", "")
WGRAD_SIDE_STREAM_3D = _ws != "0"
WGRAD_SIDE_STREAM_2D = _ws != "0"

# Forward: the side branches of a decoder stage -- projection / predictor heads and the deep-supervision map (pcrlv2_model_3d.py:67-71), small
# HBM- and latency-bound kernels whose results nothing in the forward consumes -- run on the side stream next to the next stage's convolutions;
# the main stream joins at the end of model.forward (train_3d.step_losses: once, before the losses).  PCRL_BRANCH_STREAM=0: off (A/B switch;
# results are bit-identical).
FWD_BRANCH_STREAM = os.environ.get("PCRL_BRANCH_STREAM", "1") != "0"

# The same for the 2D path (train_2d.step_losses: view 2 next to view 1; the global cosine term sits between the second view and the local
# views, so the join comes before it).  PCRL_VIEW_STREAMS_2D=0: off.
VIEW_STREAMS_2D = os.environ.get("PCRL_VIEW_STREAMS_2D", "1") != "0"

# Composed up-conv, bias gradients: the border-class sums of dy0 (the gradient entering conv1 = the output of its training-mode BatchNorm's
# backward, whose per-channel sum over the batch is zero in exact arithmetic) read only the border voxels; the interior class is minus the
# rest (pcrl_upconv_wgrad_accum flags bit 1).  config.UPC_ZERO_SUM = False (the tests' A/B handle; no environment switch since round 6): sum every voxel (A/B switch; differs by the rounding of dy0).
UPC_ZERO_SUM = True

# Composed up-conv: the chain rule from the accumulated gradient of the composed weights to the reference parameters (weight-sized GEMMs and
# re-layouts, ~0.3 ms per stage) is queued on the side stream as soon as the stage's LAST backward pass has accumulated (the step's first
# forward pass) -- next to the rest of the backward -- instead of on the main stream at the end of backward(), where it was a serial tail on
# an idle chip.  config.EARLY_COMPOSED = False (the tests' A/B handle; no environment switch since round 6): at the end of backward(), on the main stream (A/B switch; bit-identical).
EARLY_COMPOSED = True

# The host enqueues a step in ~14 ms, the GPU runs it in ~34: left alone the host runs as far ahead as the launch queue lets it, and every
# block whose last use was recorded on another stream (the side / view streams: record_stream) stays unavailable to the caching allocator
# until the GPU gets there -- the reserved footprint grew to 8x the peak allocation (97 GB for 12.6 GB at b = 32) with device mallocs inside
# the timed steps.  train_step therefore waits, before enqueueing step k, for the END of step k - MAX_STEPS_AHEAD (an event, not a device
# synchronize): the GPU always has at least one whole step queued, the footprint stays at a few times the peak.  0: unlimited.
MAX_STEPS_AHEAD = int(os.environ.get("PCRL_MAX_STEPS_AHEAD", "2"))

# Forward + backward of the SECOND global view on its own stream (train_3d.step_losses): the two global views share nothing but the
# parameters, the packed-weight caches (built by the first view: guarded by an event) and the BatchNorm running statistics (updated in the
# reference's order: the second view's update of a layer waits for the first view's, ops.order_rmw).  HBM-bound passes and launch gaps of one
# view then run under the other's MFMA work.  Needs the weight-gradient side stream (the composed up-conv's accumulators are ordered there).
# Same-box A/B 36.0 -> 34.67 ms (+3.8 %).  PCRL_VIEW_STREAMS=0: off (A/B switch; results are bit-identical).
VIEW_STREAMS = os.environ.get("PCRL_VIEW_STREAMS", "1") != "0"

# The global-average-pool branch of UpTransition (pcrlv2_model_3d.py:67) sends d_g[n][c] / S back to every voxel of a1: folded into the
# two passes of ops.1's BatchNorm backward (pcrl_bn_act_bwd_*_rowadd) instead of materialised (pcrl_gap_bwd).  config.FOLD_GAP_GRAD = False (the tests' A/B handle; no environment switch since round 6):
# materialise (A/B switch; the folded form skips one bf16 rounding of the summed gradient).
FOLD_GAP_GRAD = True

# The second LUConv of an encoder stage and the MaxPool3d(2) behind it run as one autograd node whose backward folds
# max_pool3d_backward into the BatchNorm backward passes (functions.LUConvPoolFn, pcrl_bn_act_bwd_*_pool).  config.FOLD_POOL_GRAD = False (the tests' A/B handle; no environment switch since round 6): the
# separate nodes (A/B switch).
FOLD_POOL_GRAD = True

# Forward: the normalise+activate pass of a LUConv also produces what its only other consumer needs -- MaxPool3d(2) of the result (encoder
# stage end) or its global average pool (UpTransition) -- instead of a second kernel re-reading the activation it just wrote
# (pcrl_bn_act_apply_pool / pcrl_bn_act_apply_gap).  config.FUSE_APPLY_CONSUMERS = False (the tests' A/B handle; no environment switch since round 6): separate kernels (A/B switch; results are bit-identical).
FUSE_APPLY_CONSUMERS = True
# The data gradient of ops.1 takes the first pass of ops.0's BatchNorm backward from its own output tiles (pcrl_conv3d_k3_dgrad_bnred: wide-brick bf16 shapes
# behind ReLU; ops.luconv_backward's `bnred`).  PCRL_DGRAD_BNRED=0: the separate reduce pass (A/B switch; same arithmetic per element, different summation order).
DGRAD_BNRED = os.environ.get("PCRL_DGRAD_BNRED", "1") != "0"

# UpTransition: up_conv (ConvTranspose3d k2 s2) and ops.0.conv1 (3x3x3) are applied back to back (pcrlv2_model_3d.py:64) -- composed into
# one 8-tap operator on the coarse grid (csrc/upconv_fused.hip): 0.30 of the multiply-adds of the 27-tap convolution over the upsampled
# tensor, which is never formed.  Training-mode BatchNorm route only.  PCRL_COMPOSE_UPCONV=0: the two separate kernels (A/B switch).
COMPOSE_UPCONV = os.environ.get("PCRL_COMPOSE_UPCONV", "1") != "0"

# Allocator provisioning (ops.provision_allocator): after the first complete training step the per-stream pools of torch's caching
# allocator are grown to PROVISION_FACTOR times what that step left in them, once, so that the multi-stream steady state (blocks in flight
# across the two-step run-ahead window) needs no hipMalloc later -- the timed region of a short benchmark run (`--warmup 5`) otherwise holds
# a few device mallocs per step.  1: off.  Round 4: 2 instead of 3 -- measured on MI355X with the driver's `--warmup 5`-style run: still 0 device
# mallocs in the timed region, 28.8 GB reserved instead of 44.7 for the 13 GB peak of a C2 step (sized to what the pools need).
PROVISION_FACTOR = int(os.environ.get("PCRL_PROVISION_FACTOR", "2"))

# (Round 3's experiment -- the three forwards enqueued stage by stage in rotation, each on its own stream, with one large matrix kernel at a time
# across streams -- measured slower in every combination, 33.5 -> 34.4 / 37.1 / 38.5 ms, and is gone from the code; DESIGN section 5.4 keeps the result.)

# Weight gradients of the second view's backward on the view stream itself (ops.side_wgrad) instead of the side stream.  Measured without a
# profiler (tools/stream_balance_probe.py: events at the tail of every stream): the view stream ran dry 24.2 ms into a 31.6 ms step -- it
# carries one pass, the main stream two (view 1 + local views) and the side stream the weight gradients of all three -- so for a quarter of
# the step only two streams fed the chip.  With the second view's plain weight gradients inline (the composed up-conv's accumulations stay on
# the side stream: they add into buffers the passes share, ordered by that stream) the view stream ends at 29.4 ms and the step goes
# 32.36 -> 31.84 ms (same box, three interleaved runs each).  config.VIEW_WGRAD_INLINE = False (the tests' A/B handle; no environment switch since round 6): off (A/B switch; results are bit-identical).
VIEW_WGRAD_INLINE = True

# (Measured, not kept, removed in round 5: the second view's side branches on its own stream; a start skew between the two views' forwards; HIP stream
# priorities for the view / side streams -- 31.80 / 31.81 / 31.82 / 31.80 ms, they do nothing.)

# Weight packing off the forward's critical chain: the packed / composed weight forms a step needs (10 pack launches, 3 x the composed
# operator's prep + GEMM + pack + bias: ~25 small kernels, 0.5 ms back to back) are rebuilt on the SIDE stream at the start of the step, in
# first-use order, while the main stream already runs the first layers; every reader waits for its own cache's event (ops._CacheGuard).
# Without it each pack launch sits in front of its convolution on the first view's chain and the second view waits for it too.  The first
# step of a model records which caches it builds (ops.prepack).  config.PREPACK = False (the tests' A/B handle; no environment switch since round 6): off (A/B switch; results are bit-identical).
PREPACK = True

# The passes' parameter gradients summed into the optimizer's arena by ONE launch of our own (pcrl_grad_sum) instead of a multi-tensor copy
# and two multi-tensor adds (four ATen launches on the serial tail of backward).  config.FUSED_GRAD_SUM = False (the tests' A/B handle; no environment switch since round 6): off (bit-identical).
FUSED_GRAD_SUM = True

# The reference returns the caching allocator's cached memory to the driver after every epoch (train_3d.py:83, "help release GPU memory").  It
# changes no result.  On by default (as the reference), through ops.empty_cache: the provisioned per-stream pools of the steady state are held
# across the call (round 3 had to switch the call off: releasing and re-reserving 45 GB of pools stalled ~3 s one epoch in ten) -- everything
# else the allocator caches goes back to the driver.  PCRL_EMPTY_CACHE_PER_EPOCH=0: no call; =raw: torch.cuda.empty_cache() itself.
_ec = os.environ.get("PCRL_EMPTY_CACHE_PER_EPOCH", "1")
EMPTY_CACHE_PER_EPOCH = _ec != "0"
EMPTY_CACHE_RAW = _ec == "raw"
