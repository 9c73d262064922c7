"""Global knobs of the MI355X engine."""
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
# rest (pcrl_upconv_wgrad_accum flags bit 1).  PCRL_UPC_ZERO_SUM=0: sum every voxel (A/B switch; differs by the rounding of dy0).
UPC_ZERO_SUM = os.environ.get("PCRL_UPC_ZERO_SUM", "1") != "0"

# Composed up-conv: the chain rule from the accumulated gradient of the composed weights to the reference parameters (weight-sized GEMMs and
# re-layouts, ~0.3 ms per stage) is queued on the side stream as soon as the stage's LAST backward pass has accumulated (the step's first
# forward pass) -- next to the rest of the backward -- instead of on the main stream at the end of backward(), where it was a serial tail on
# an idle chip.  PCRL_EARLY_COMPOSED=0: at the end of backward(), on the main stream (A/B switch; bit-identical).
EARLY_COMPOSED = os.environ.get("PCRL_EARLY_COMPOSED", "1") != "0"

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
# two passes of ops.1's BatchNorm backward (pcrl_bn_act_bwd_*_rowadd) instead of materialised (pcrl_gap_bwd).  PCRL_FOLD_GAP_GRAD=0:
# materialise (A/B switch; the folded form skips one bf16 rounding of the summed gradient).
FOLD_GAP_GRAD = os.environ.get("PCRL_FOLD_GAP_GRAD", "1") != "0"

# The second LUConv of an encoder stage and the MaxPool3d(2) behind it run as one autograd node whose backward folds
# max_pool3d_backward into the BatchNorm backward passes (functions.LUConvPoolFn, pcrl_bn_act_bwd_*_pool).  PCRL_FOLD_POOL_GRAD=0: the
# separate nodes (A/B switch).
FOLD_POOL_GRAD = os.environ.get("PCRL_FOLD_POOL_GRAD", "1") != "0"

# Forward: the normalise+activate pass of a LUConv also produces what its only other consumer needs -- MaxPool3d(2) of the result (encoder
# stage end) or its global average pool (UpTransition) -- instead of a second kernel re-reading the activation it just wrote
# (pcrl_bn_act_apply_pool / pcrl_bn_act_apply_gap).  PCRL_FUSE_APPLY_CONSUMERS=0: separate kernels (A/B switch; results are bit-identical).
FUSE_APPLY_CONSUMERS = os.environ.get("PCRL_FUSE_APPLY_CONSUMERS", "1") != "0"

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

# EXPERIMENT, measured and NOT kept (default off; kept as switches because the result is instructive -- DESIGN.md section 5).
# Idea: two streams that run the SAME layer sequence from the same start fall into lockstep -- both convolutions side by side, then both
# BatchNorm passes side by side -- so HBM-bound passes never run under the other pass's matrix work (rocprofv3 timeline: 7.6 ms per step with
# only HBM-bound kernels in flight).  INTERLEAVE_VIEWS: the three forwards are enqueued stage by stage in rotation (PCRLv23d.forward_views:
# view 1, view 2, local views, each on its own stream), so the backward replays in the same rotation.  MFMA_TOKEN: every large matrix kernel
# (>= MFMA_TOKEN_MIN_GF GFLOP) waits for the previous large matrix kernel of another stream (ops.mfma_turn): one matrix kernel at a time, in
# enqueue order, HBM-bound passes under it.  Results bit-identical (tests).  Same-box A/B, 15 steps x 2 rounds: baseline 33.53 ms; interleave
# alone 34.39; interleave + token 37.13; token alone 38.54; token only >= 60 GFLOP 35.59.  Why: (a) a cross-stream event wait in front of
# ~150 kernels per step exposes the queue-to-queue signalling latency each time; (b) two matrix kernels side by side are ~10 % FASTER than
# back to back (the co-scheduled blocks fill each other's barrier / staging stalls: one kernel alone keeps the MFMA pipe 56 % busy), so
# serialising them gives up more than the hidden BatchNorm passes return; (c) step time tracks the SUM of kernel time (a step whose
# draws leave view 2's 64-channel scale without a gradient skips 3.4 ms of kernels and is 3 ms shorter), i.e. the chip is throughput-bound.
INTERLEAVE_VIEWS = os.environ.get("PCRL_INTERLEAVE", "0") == "1"
MFMA_TOKEN = os.environ.get("PCRL_MFMA_TOKEN", "0") == "1"
MFMA_TOKEN_MIN_GF = float(os.environ.get("PCRL_MFMA_TOKEN_MIN_GF", "20"))

# Weight gradients of the second view's backward on the view stream itself (ops.side_wgrad) instead of the side stream.  Measured without a
# profiler (tools/stream_balance_probe.py: events at the tail of every stream): the view stream ran dry 24.2 ms into a 31.6 ms step -- it
# carries one pass, the main stream two (view 1 + local views) and the side stream the weight gradients of all three -- so for a quarter of
# the step only two streams fed the chip.  With the second view's plain weight gradients inline (the composed up-conv's accumulations stay on
# the side stream: they add into buffers the passes share, ordered by that stream) the view stream ends at 29.4 ms and the step goes
# 32.36 -> 31.84 ms (same box, three interleaved runs each).  PCRL_VIEW_WGRAD_INLINE=0: off (A/B switch; results are bit-identical).
VIEW_WGRAD_INLINE = os.environ.get("PCRL_VIEW_WGRAD_INLINE", "1") != "0"

# EXPERIMENT: the second view's forward side branches (heads, deep-supervision map) on the view stream itself instead of the side stream
# (the view stream is idle for the last 2 ms of the forward phase).  PCRL_VIEW_BRANCH_INLINE=1: on.
VIEW_BRANCH_INLINE = os.environ.get("PCRL_VIEW_BRANCH_INLINE", "0") == "1"

# Experiment: the second view's forward starts only when the first view's forward has passed its VIEW_SKEW-th stage stop (1..11; 0: off --
# both views start together and run their identical layer sequences in lockstep, convolution next to convolution and BatchNorm next to
# BatchNorm).  One event wait per step, nothing else changes; results bit-identical.
VIEW_SKEW = int(os.environ.get("PCRL_VIEW_SKEW", "0"))

# Experiment: HIP stream priorities (0 = normal, -1 = high).  When two streams hold ready kernels the dispatcher takes the higher-priority
# queue's workgroups first, so a favoured view finishes its convolution earlier and its BatchNorm passes run under the other view's
# convolution instead of next to its BatchNorm passes (the two views run identical layer sequences).  Results bit-identical.
VIEW_STREAM_PRIORITY = int(os.environ.get("PCRL_VIEW_PRIO", "0"))
SIDE_STREAM_PRIORITY = int(os.environ.get("PCRL_SIDE_PRIO", "0"))

# Weight packing off the forward's critical chain: the packed / composed weight forms a step needs (10 pack launches, 3 x the composed
# operator's prep + GEMM + pack + bias: ~25 small kernels, 0.5 ms back to back) are rebuilt on the SIDE stream at the start of the step, in
# first-use order, while the main stream already runs the first layers; every reader waits for its own cache's event (ops._CacheGuard).
# Without it each pack launch sits in front of its convolution on the first view's chain and the second view waits for it too.  The first
# step of a model records which caches it builds (ops.prepack).  PCRL_PREPACK=0: off (A/B switch; results are bit-identical).
PREPACK = os.environ.get("PCRL_PREPACK", "1") != "0"

# The passes' parameter gradients summed into the optimizer's arena by ONE launch of our own (pcrl_grad_sum) instead of a multi-tensor copy
# and two multi-tensor adds (four ATen launches on the serial tail of backward).  PCRL_FUSED_GRAD_SUM=0: off (bit-identical).
FUSED_GRAD_SUM = os.environ.get("PCRL_FUSED_GRAD_SUM", "1") != "0"

# The reference returns the caching allocator's cached memory to the driver after every epoch (train_3d.py:83, "help release GPU memory").  It
# changes no result.  On by default (as the reference), through ops.empty_cache: the provisioned per-stream pools of the steady state are held
# across the call (round 3 had to switch the call off: releasing and re-reserving 45 GB of pools stalled ~3 s one epoch in ten) -- everything
# else the allocator caches goes back to the driver.  PCRL_EMPTY_CACHE_PER_EPOCH=0: no call; =raw: torch.cuda.empty_cache() itself.
_ec = os.environ.get("PCRL_EMPTY_CACHE_PER_EPOCH", "1")
EMPTY_CACHE_PER_EPOCH = _ec != "0"
EMPTY_CACHE_RAW = _ec == "raw"

# Experiment: a layer's weight gradient (side stream) is queued BEHIND its data gradient instead of in front of it.  Both need the same dy;
# queued first, the weight gradient runs next to the data gradient (two matrix kernels sharing the chip) and the BatchNorm backward passes
# of the layer below then run with nothing beside them; queued second, the side stream's wait covers the data gradient, and the weight
# gradient runs next to those HBM-bound passes.  Same kernels, same results (bit-identical).  PCRL_WGRAD_AFTER_DGRAD=1: on.
WGRAD_AFTER_DGRAD = os.environ.get("PCRL_WGRAD_AFTER_DGRAD", "0") == "1"

# Experiment (measured, no gain, default off): weight gradients of the 1-output-channel layers (OutputTransition, the deep-supervision heads) on
# the side stream like every other weight gradient instead of inline on the data-gradient chain, at whose head OutputTransition's column sums
# are ~100 us.  Same box, 4 interleaved runs: 31.70 (inline) vs 31.80 ms.  PCRL_TO1_WGRAD_SIDE=1: on (bit-identical).
TO1_WGRAD_SIDE = os.environ.get("PCRL_TO1_WGRAD_SIDE", "0") == "1"
