"""Host-side functional layer over libpcrl_hip.so (plain functions on device tensors, no autograd).

Tensor conventions (see include/pcrl_hip.h):
  * activations: torch tensors of logical shape [N, C, D, H, W] whose MEMORY is NDHWC (torch's
    channels_last_3d), dtype float32 or bfloat16 -- `new_act` / `dims` below;
  * 1-channel maps, head tensors [rows, C], parameters and their gradients: contiguous float32.
Every function launches on torch's current stream and returns immediately.
"""
from __future__ import annotations

import collections
import os
import weakref

import torch

from . import config
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, CONV_BM, PcrlError, dtype_code, lib, stream_handle

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# ----------------------------------------------------------------------------------------------
# memory helpers
# ----------------------------------------------------------------------------------------------
_ws_cache: dict = {}


def workspace(nbytes: int, device, arena=None) -> torch.Tensor:
    """Grow-only scratch arena per device AND stream (the launches of one stream are ordered, so they can share one; a second
    stream -- ops.side_wgrad -- gets its own)."""
    if arena is None:
        arena = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device.type, device.index, arena)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


# ----------------------------------------------------------------------------------------------
# roctx ranges (PCRL_TRACE_RANGES=1): forward / backward / optimizer / bucket all-reduce show up as named ranges in a
# `rocprofv3 --kernel-trace --marker-trace` run (the reference has only wall-clock BT/DT meters, train_3d.py:102-103).  Off by default:
# a range costs two library calls per use.
# ----------------------------------------------------------------------------------------------
TRACE_RANGES = os.environ.get("PCRL_TRACE_RANGES", "0") == "1"
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        import ctypes
        _roctx = False
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):      # the SDK's library is the one rocprofv3 listens to
            try:
                lib_ = ctypes.CDLL(name)
                lib_.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib_.roctxRangePushA.restype = ctypes.c_int
                lib_.roctxRangePop.restype = ctypes.c_int
                _roctx = lib_
                break
            except OSError:
                continue
    return _roctx


class trace_range:
    """`with trace_range("backward"):` -- a roctx range around the enqueue of a phase (no-op unless PCRL_TRACE_RANGES=1)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = TRACE_RANGES and _roctx_lib()
        if self.on:
            _roctx.roctxRangePushA(("pcrl:" + self.name).encode())

    def __exit__(self, *exc):
        if self.on:
            _roctx.roctxRangePop()
        return False


# ----------------------------------------------------------------------------------------------
# side stream for the weight gradients
# ----------------------------------------------------------------------------------------------
# In backward a layer's weight gradient (MFMA-bound) has no consumer until the optimizer step, while the chain that everything else
# waits for -- data gradient -> BatchNorm backward of the layer below (HBM-bound) -> ... -- does not depend on it.  The weight
# gradients are therefore launched on a second stream: the matrix pipes work on them while the BatchNorm passes stream through HBM.
# Ordering: the side stream waits for the main stream before every launch (its operands were just produced there); the main stream
# waits for the side stream once, before the parked parameter gradients are summed (functions.flush_param_grads) -- operands are
# `record_stream`-ed so the caching allocator does not recycle them under the side stream; the side stream has its own scratch arena.
_side_streams: dict = {}
_side_pending: dict = {}


class side_wgrad:
    """`with side_wgrad(device, x, dy) as ws:` -- launches inside run on the side stream; ws(nbytes) is its scratch arena."""

    def __init__(self, device, *operands, path2d=False, shared_accumulator=False):
        self.device, self.operands = device, operands
        self.active = (config.WGRAD_SIDE_STREAM_2D if path2d else config.WGRAD_SIDE_STREAM_3D) and device.type == "cuda"
        if self.active and config.VIEW_WGRAD_INLINE and not shared_accumulator and not path2d and _views_active:
            # config.VIEW_WGRAD_INLINE: a weight gradient of the SECOND VIEW's backward stays on the view stream (which otherwise runs dry a
            # quarter of the step before the others) instead of queueing behind everybody's on the side stream.  Not for launches that add
            # into an accumulator the passes share (the composed up-conv's: those are ordered by the side stream).
            cur = torch.cuda.current_stream(device).cuda_stream
            if any(cur == vs.cuda_stream for vs in _view_streams.values()):
                self.active = False

    def __enter__(self):
        if not self.active:
            return lambda nb: workspace(nb, self.device)
        key = (self.device.type, self.device.index)
        side = _side_streams.get(key)
        if side is None:
            side = _side_streams[key] = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        for t in self.operands:
            t.record_stream(side)
        _side_pending[key] = True
        self._cm = torch.cuda.stream(side)
        self._cm.__enter__()
        return lambda nb: workspace(nb, self.device)      # keyed by the (now current) side stream

    def __exit__(self, *exc):
        if self.active:
            self._cm.__exit__(*exc)
        return False


class side_branch(side_wgrad):
    """`with side_branch(device, a, g):` -- a forward side branch (config.FWD_BRANCH_STREAM) on the same side stream; joined by
    join_side_stream() at the end of the forward (or, inside `deferred_join()`, when the caller asks)."""

    def __init__(self, device, *operands):
        self.device, self.operands = device, operands
        self.active = config.FWD_BRANCH_STREAM and device.type == "cuda"


_defer_join = 0


class deferred_join:
    """Inside this context model.forward leaves the side stream un-joined (its branch outputs are NOT safe to read on the main stream);
    leaving the context joins.  train_3d.step_losses wraps its three forwards in it: a stage's branch then also runs under the next pass."""

    def __enter__(self):
        global _defer_join
        _defer_join += 1

    def __exit__(self, *exc):
        global _defer_join
        _defer_join -= 1
        if _defer_join == 0:
            join_side_stream()
        return False


def end_of_forward_join():
    if _defer_join == 0:
        join_side_stream()


def join_side_stream(device=None):
    """The current stream waits for everything launched so far on the side stream and on the second view's stream (no-op when nothing is
    outstanding)."""
    for key, pending in list(_side_pending.items()):
        if pending and (device is None or (device.type, device.index) == key):
            dev = torch.device(key[0], key[1])
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(_side_streams[key])
            # "nothing outstanding" is a statement about the MAIN stream: a join issued while a view stream is current (a forward of the second
            # view outside deferred_join) makes that stream wait, but the main stream still has to
            if not any(cur.cuda_stream == vs.cuda_stream for vs in _view_streams.values()):
                _side_pending[key] = False
    if _views_active:
        # the second view's autograd nodes run their backward on the view stream without passing through view_pass: while a step uses the
        # stream every join waits for it (a wait on an idle stream costs nothing)
        for (dt_, di_, _name), vs in _view_streams.items():
            if device is None or (device.type, device.index) == (dt_, di_):
                cur = torch.cuda.current_stream(torch.device(dt_, di_))
                if cur.cuda_stream != vs.cuda_stream:
                    cur.wait_stream(vs)


def side_stream(device):
    """The side stream of `device` (created on first use): weight gradients, forward side branches -- and, under data parallelism, the
    bucket sums and collectives (ddp.DataParallel), so that a rank never drives more than four streams (ROCm's default number of
    hardware queues per process: a fifth stream shares a queue with another one and serialises with it)."""
    key = (device.type, device.index)
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=device)
    return side


def producer_streams(device):
    """The streams other than the current one that a step launches gradient producers on (side stream, view streams) -- what a consumer on
    yet another stream (the data-parallel wrapper's communication stream) has to wait for."""
    out = []
    s = _side_streams.get((device.type, device.index))
    if s is not None:
        out.append(s)
    for (dt_, di_, _name), vs in _view_streams.items():
        if (dt_, di_) == (device.type, device.index):
            out.append(vs)
    return out


# ---- host run-ahead (config.MAX_STEPS_AHEAD) ----
_step_ends: dict = {}      # device index -> deque of events recorded at the end of the last steps


def throttle_host(device, step_done=False):
    """config.MAX_STEPS_AHEAD: keep the host at most that many steps ahead of the GPU.  Called with step_done=True after optimizer.step()
    (records the step's end) and with False before a step is enqueued (waits for the end of the step MAX_STEPS_AHEAD back)."""
    lag = config.MAX_STEPS_AHEAD
    if lag <= 0 or device.type != "cuda":
        return
    q = _step_ends.setdefault(device.index, collections.deque())
    if step_done:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        q.append(ev)
    else:
        while len(q) >= lag:
            q.popleft().synchronize()


# ---- allocator provisioning (config.PROVISION_FACTOR) ----
# The step runs on three streams and the host runs up to MAX_STEPS_AHEAD steps ahead of the GPU.  torch's caching allocator keeps a pool PER
# STREAM, and a block whose last use was recorded on another stream (record_stream: the operands of side-stream weight gradients, the second
# view's inputs) only returns to its pool when the GPU has got there -- up to two steps later.  Left alone the pools grow by a few hipMalloc
# per step (each a device-wide stall) for the first 15-25 steps until every stream's pool holds enough blocks for the whole run-ahead window:
# with the driver's `--warmup 5` that growth was inside the timed region (23 mallocs in 20 steps, 31-36 ms step times).  Instead the engine
# sizes the pools ONCE, after the first complete step of a (model, batch shape): every segment the first step left in a stream's pool is
# duplicated PROVISION_FACTOR - 1 times on the same stream (288 GB of HBM: the 3x of a 13 GB peak is nothing), so the steady state is there
# from step 2.  One device synchronisation, once.
_provisioned: dict = {}                # (device index, key) -> bytes reserved right after the pools were sized for that key
_provisioned_segments: set = set()     # (device, address) of the segments that existed after the last provisioning: only NEW ones are duplicated


def provision_allocator(device, key=None):
    f = config.PROVISION_FACTOR
    if f <= 1 or device.type != "cuda":
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    done = _provisioned.get((idx, key))
    if done is not None:
        if torch.cuda.memory_reserved(device) >= 0.7 * done:      # a host-side counter: no synchronisation
            return
        # the pools were emptied since (torch.cuda.empty_cache() at the end of an epoch, train_3d.py:83): size them again
        for k in [k for k in _provisioned if k[0] == idx]:
            del _provisioned[k]
        _provisioned_segments.difference_update({e for e in _provisioned_segments if e[0] == idx})
    _provisioned[(idx, key)] = 0
    torch.cuda.synchronize(device)
    import time as _time
    _t0 = _time.perf_counter()
    segs = [s for s in torch.cuda.memory_snapshot() if s["device"] == idx and (idx, s["address"]) not in _provisioned_segments]
    free_b, total_b = torch.cuda.mem_get_info(device)
    want = sum(s["total_size"] for s in segs) * (f - 1)
    if want > 0.5 * free_b:           # never provision a box into memory pressure: scale the copies down
        f = 1 + int((f - 1) * 0.5 * free_b / max(want, 1))
        if f <= 1:
            return
    streams = {}
    for st in [torch.cuda.current_stream(device)] + list(_side_streams.values()) + list(_view_streams.values()):
        streams[st.cuda_stream] = st
    hold = []
    by_stream = collections.defaultdict(list)
    for s in segs:
        by_stream[s["stream"]].append(s)
    for sid, lst in by_stream.items():
        st = streams.get(sid)
        if st is None:          # the null stream / a stream this module does not own (RCCL, data-parallel wrapper): leave its pool alone
            continue
        with torch.cuda.stream(st):
            # (1) occupy what the pool holds now, so that (2) has to come from the device
            for s in lst:
                for blk in s["blocks"]:
                    if blk["state"] == "inactive" and blk["size"] >= 512:
                        try:
                            hold.append(torch.empty(blk["size"], dtype=torch.uint8, device=device))
                        except RuntimeError:
                            pass
            # (2) f - 1 more segments of every size.  Small-pool segments (2 MB) are filled with two 1 MB - 512 B requests each.
            for s in lst:
                for _ in range(f - 1):
                    try:
                        if s["segment_type"] == "small":
                            hold.append(torch.empty((1 << 20) - 512, dtype=torch.uint8, device=device))
                            hold.append(torch.empty((1 << 20) - 512, dtype=torch.uint8, device=device))
                        else:
                            hold.append(torch.empty(s["total_size"], dtype=torch.uint8, device=device))
                    except RuntimeError:
                        break
    del hold
    torch.cuda.synchronize(device)
    for s in torch.cuda.memory_snapshot():
        if s["device"] == idx:
            _provisioned_segments.add((idx, s["address"]))
    _provisioned[(idx, key)] = torch.cuda.memory_reserved(device)
    if os.environ.get("PCRL_PROVISION_VERBOSE", "0") == "1":
        print("[provision] %d new segments x%d -> reserved %.1f GB in %.2f s" % (len(segs), f, _provisioned[(idx, key)] / 2**30, _time.perf_counter() - _t0), flush=True)


def empty_cache(device=None):
    """The reference's per-epoch `torch.cuda.empty_cache()` (train_3d.py:83, train_2d.py:108: "help release GPU memory") WITHOUT giving up the
    engine's provisioned per-stream pools: every inactive block of the segments provision_allocator sized is occupied by a placeholder for
    the duration of the call, so the allocator returns to the driver everything else it caches (other streams' and other shapes' leftovers:
    what the call is for) and keeps the steady-state pools -- releasing and re-reserving those cost one device-wide stall of ~3 s in ten
    epochs (round 3, which therefore switched the call off).  One device synchronisation per epoch, like the reference's own."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    idx = device.index if device.index is not None else torch.cuda.current_device()
    mine = {e[1] for e in _provisioned_segments if e[0] == idx}
    if not mine:
        torch.cuda.empty_cache()
        return
    torch.cuda.synchronize(device)
    streams = {}
    for st in [torch.cuda.current_stream(device)] + list(_side_streams.values()) + list(_view_streams.values()):
        streams[st.cuda_stream] = st
    # blocks freed while another stream still used them (record_stream) sit in the allocator as "active_pending_free" until it processes their
    # events -- which it does at the start of the NEXT malloc, not at the synchronize above.  One throw-away allocation per stream settles them into
    # "inactive" before the snapshot is read (left pending, they turned free under the placeholders and took one meant for a provisioned segment)
    for st_ in streams.values():
        with torch.cuda.stream(st_):
            torch.empty(1, dtype=torch.uint8, device=device)
    hold = []
    by_stream = collections.defaultdict(list)
    seen = {}
    for seg in torch.cuda.memory_snapshot():
        if seg["device"] == idx and seg["address"] in mine:
            seen[seg["address"]] = (seg.get("segment_type", "?"), seg["total_size"], seg["stream"] in streams,
                                    [(b_["state"], b_["size"]) for b_ in seg["blocks"]][:6])
        if seg["device"] == idx and seg["stream"] in streams:
            # EVERY inactive block of the step's streams gets a placeholder, whichever segment it sits in: best fit gives a request the smallest free
            # block that holds it, so with the requests issued largest first each one takes a block of exactly its size -- but WHICH of several equal
            # (or larger, unclaimed) blocks is the allocator's choice, and a placeholder sized from a provisioned segment's hole that lands in a hole
            # of another segment leaves the provisioned one to be released (seen in normal runs: one wholly free 16 MiB segment per call).  With every
            # hole claimed, the placeholders are sorted by where they LANDED: those outside the provisioned segments are dropped before the cache is
            # emptied (that memory goes back to the driver, as the reference's empty_cache wants), those inside are held across it.
            # A request lands in the pool its SIZE selects (torch's caching allocator: <= 1 MiB -> the small pool of 2 MiB segments, above -> the large
            # pool), whatever segment the hole it was sized from belongs to.  So a placeholder can only pin a hole of its own pool (filtering
            # by block size alone let a >= 1 MiB hole of a small-pool segment ask the LARGE pool for a block -- taking one meant for another
            # placeholder or opening a fresh 20 MiB segment, and the "segments were released" warning fired in normal runs):
            #   large-pool segment: holes above 1 MiB (smaller holes cannot be held: a request of that size is a small-pool request)
            #   small-pool segment: holes of at most 1 MiB, held by a request of that size; a larger hole (a 2 MiB segment that is mostly free) by
            #                       1 MiB requests -- best fit places them in it
            large = seg.get("segment_type", "large") == "large"
            for blk in seg["blocks"]:
                if blk["state"] != "inactive":
                    continue
                if large and blk["size"] > (1 << 20):
                    by_stream[seg["stream"]].append(blk["size"])
                elif not large and blk["size"] >= 512:
                    n_, left_ = blk["size"], []
                    while n_ > (1 << 20):
                        left_.append(1 << 20)
                        n_ -= 1 << 20
                    if n_ >= 512:
                        left_.append(n_)
                    by_stream[seg["stream"]].extend(left_)
    for sid, sizes in by_stream.items():
        with torch.cuda.stream(streams[sid]):
            for n in sorted(sizes, reverse=True):      # largest first: best fit then takes exactly the block the request was sized from
                try:
                    hold.append(torch.empty(n, dtype=torch.uint8, device=device))
                except RuntimeError:
                    break
    ranges = sorted((a_, a_ + v_[1]) for a_, v_ in seen.items())
    import bisect
    starts = [r_[0] for r_ in ranges]

    def inside(t):
        q = t.data_ptr()
        k = bisect.bisect_right(starts, q) - 1
        return k >= 0 and q < ranges[k][1]
    hold = [t for t in hold if inside(t)]          # the others are freed here: their (non-provisioned) segments may go
    torch.cuda.empty_cache()
    del hold
    # a provisioned segment can still be lost (a hole nothing could hold: under 512 bytes, or under 1 MiB in a large-pool segment): say so once --
    # the next step re-reserves it (a device-wide stall), which is what this function exists to avoid
    left = {seg["address"] for seg in torch.cuda.memory_snapshot() if seg["device"] == idx}
    lost = [a for a in mine if a not in left]
    if lost:
        for e in list(_provisioned_segments):
            if e[0] == idx and e[1] in lost:
                _provisioned_segments.discard(e)
        global _empty_cache_warned
        if not _empty_cache_warned:
            _empty_cache_warned = True
            print(f"[pcrlv2_amd.ops] empty_cache: {len(lost)} of {len(mine)} provisioned segments were released with the cache "
                  "(a placeholder landed elsewhere); they are re-reserved on demand", flush=True)
            if os.environ.get("PCRL_PROVISION_VERBOSE", "0") == "1":
                for a_ in lost:
                    print("    lost segment (type, bytes, on a known stream, first blocks):", seen.get(a_), flush=True)


_empty_cache_warned = False


# ---- the second global view on its own stream (config.VIEW_STREAMS) ----
_view_streams: dict = {}
# the passes that get a stream of their own: the second global view (the local views' pass stays on the main stream -- a third view stream was
# measured and is slower: 31.64 -> 32.02 ms)
VIEW_STREAM_NAMES = ("view2",)
_views_active = False        # set while a step uses the view stream: the cross-stream guards below are then live
_rmw_events: dict = {}


def view_streams_on(device, path2d=False) -> bool:
    if path2d:      # the 2D path has no accumulators shared between passes: no need for the side stream
        return config.VIEW_STREAMS_2D and device.type == "cuda"
    return config.VIEW_STREAMS and config.WGRAD_SIDE_STREAM_3D and device.type == "cuda"


def fork_views(device, path2d=False):
    """Start of a step, BEFORE the first view is queued: the view stream waits for what the main stream holds now (the optimizer step)."""
    global _views_active
    if not view_streams_on(device, path2d):
        return
    for name in VIEW_STREAM_NAMES:
        key = (device.type, device.index, name)
        vs = _view_streams.get(key)
        if vs is None:
            vs = _view_streams[key] = torch.cuda.Stream(device=device)
        vs.wait_stream(torch.cuda.current_stream(device))
    _views_active = True


def view_stream(device, name, path2d=False):
    """The stream of pass `name` while a step uses view streams (fork_views was called), else None (= the current stream)."""
    if not (_views_active and view_streams_on(device, path2d) and name in VIEW_STREAM_NAMES):
        return None
    return _view_streams[(device.type, device.index, name)]


class view_pass:
    """`with view_pass(device, x, name="view2"):` -- the forward queued inside runs on that view stream (its autograd nodes run their backward there)."""

    def __init__(self, device, *operands, name="view2", path2d=False):
        self.device, self.operands, self.name = device, operands, name
        self.active = _views_active and view_streams_on(device, path2d) and name in VIEW_STREAM_NAMES

    def __enter__(self):
        if self.active:
            key = (self.device.type, self.device.index, self.name)
            vs = _view_streams[key]
            for t in self.operands:
                t.record_stream(vs)
            self._cm = torch.cuda.stream(vs)
            self._cm.__enter__()

    def __exit__(self, *exc):
        if self.active:
            self._cm.__exit__(*exc)
        return False


def order_rmw(t):
    """Before a read-modify-write of a tensor the passes share (BatchNorm running statistics): wait for the last update made on another stream."""
    if not _views_active or t is None:
        return
    e = _rmw_events.get(t.data_ptr())
    if e is not None:
        cur = torch.cuda.current_stream(t.device)
        if e[1] != cur.cuda_stream:
            cur.wait_event(e[0])


def mark_rmw(t):
    if not _views_active or t is None:
        return
    cur = torch.cuda.current_stream(t.device)
    ev = torch.cuda.Event()
    ev.record(cur)
    _rmw_events[t.data_ptr()] = (ev, cur.cuda_stream)


class _CacheGuard:
    """Mixin of the packed-weight caches: built on one stream, read on another -> the reader waits for the build."""
    _ev = None
    _ev_stream = None

    def _built(self, device):
        if _views_active:
            cur = torch.cuda.current_stream(device)
            self._ev = torch.cuda.Event()
            self._ev.record(cur)
            self._ev_stream = cur.cuda_stream
        else:
            self._ev = None

    def _reading(self, device):
        if _views_active and self._ev is not None:
            cur = torch.cuda.current_stream(device)
            if cur.cuda_stream != self._ev_stream:
                cur.wait_event(self._ev)


def new_act(N, D, H, W, C, dtype, device) -> torch.Tensor:
    return torch.empty((N, D, H, W, C), dtype=dtype, device=device).permute(0, 4, 1, 2, 3)


def dims(t: torch.Tensor):
    """(N, D, H, W, C) of an NDHWC-memory activation; raises if the memory layout is not NDHWC."""
    if t.dim() != 5:
        raise PcrlError(f"expected a 5-D activation, got shape {tuple(t.shape)}")
    N, C, D, H, W = t.shape
    if not t.permute(0, 2, 3, 4, 1).is_contiguous():
        raise PcrlError("activation is not NDHWC (channels_last_3d) contiguous")
    return N, D, H, W, C


def to_act(x: torch.Tensor, dtype) -> torch.Tensor:
    """API-boundary glue: any [N,C,D,H,W] tensor -> NDHWC memory in `dtype` (no-op when already so)."""
    if x.dtype != dtype:
        x = x.to(dtype)
    if not x.permute(0, 2, 3, 4, 1).is_contiguous():
        x = x.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    return x


def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


_zero_cache: dict = {}


def zero_grad_vector(n, device):
    """Gradient of a bias that is identically zero.  With engine-delivered parameter gradients the same read-only zeros are
    handed out every time (no fill kernel; functions.flush_param_grads copies, never adopts, a shared tensor)."""
    if not config.DIRECT_PARAM_GRADS:
        return torch.zeros(n, dtype=torch.float32, device=device)
    key = (n, str(device))
    t = _zero_cache.get(key)
    if t is None:
        t = _zero_cache[key] = torch.empty(n, dtype=torch.float32, device=device)
        if t.is_cuda:
            lib().call("pcrl_zero", t, 4 * n, stream_handle())      # (first use of a size, from inside a backward: a runtime memset, no ATen launch)
        else:
            t.zero_()
        t._pcrl_shared_zero = True
    return t


def is_shared_zero(t) -> bool:
    return getattr(t, "_pcrl_shared_zero", False)


# ----------------------------------------------------------------------------------------------
# weight packing (cached per parameter version)
# ----------------------------------------------------------------------------------------------
_weights_epoch = 0


def bump_weights_epoch():
    """Called by optimizers that update parameters outside torch's version counter."""
    global _weights_epoch
    _weights_epoch += 1


# ----------------------------------------------------------------------------------------------
# forward-pass index within one optimizer step (0 = first forward after optimizer.step(): its backward runs LAST, so the
# parameter gradients it produces are final -- used by ddp.DataParallel to start all-reduces during backward)
# ----------------------------------------------------------------------------------------------
_pass_index = 0


def begin_step():
    global _pass_index, _views_active, _pack_recording
    _pass_index = 0
    _views_active = False
    _pack_recording = None
    drop_pending_composed()


def next_pass() -> int:
    global _pass_index
    i = _pass_index
    _pass_index += 1
    return i


# ---- config.PREPACK: the caches a step builds, rebuilt on the side stream at the start of the next steps ----
_pack_plans = weakref.WeakKeyDictionary()     # model -> [(cache, args of its get())] in first-use order
_pack_recording = None


def _record_build(cache, args):
    if any(getattr(a, "_pcrl_no_prepack", False) for a in args if torch.is_tensor(a)):
        return          # derived tensors refreshed inside forward (pad_first_layer): packing them at step start would pack last step's values
    if _pack_recording is not None and not any(c is cache for c, _ in _pack_recording):
        _pack_recording.append((cache, args))


def prepack(model, device):
    """Start of a step's forward (after fork_views): rebuild the weight forms the step will ask for on the side stream, ahead of their first
    use.  A model's first step records them instead (every cache miss between this call and the next begin_step())."""
    global _pack_recording
    _pack_recording = None
    if not (config.PREPACK and _views_active and device.type == "cuda"):
        return
    plan = _pack_plans.get(model)
    if plan is None:
        _pack_recording = _pack_plans[model] = []
        return
    if not plan:
        return
    ss, cur = side_stream(device), torch.cuda.current_stream(device)
    ss.wait_stream(cur)          # the optimizer step that changed the weights; every reader of the old forms finished before it
    with torch.cuda.stream(ss):
        for cache, args in plan:
            cache.get(*args)
    _side_pending[(device.type, device.index)] = True


class PackedWeights(_CacheGuard):
    """Packed (K-contiguous, activation-dtype) copies of one conv / transposed-conv weight."""

    def __init__(self, kind: str):
        self.kind = kind  # 'conv3' | 'convt'
        self.key = None
        self.fwd = None
        self.dgrad = None

    def get(self, w: torch.Tensor, dtype):
        key = (_weights_epoch, w._version, w.data_ptr(), dtype)
        if key != self.key:
            L, s = lib(), stream_handle()
            self.fwd = torch.empty(w.numel(), dtype=dtype, device=w.device)
            self.dgrad = torch.empty(w.numel(), dtype=dtype, device=w.device)
            if self.kind == "conv3":
                Co, Ci = w.shape[0], w.shape[1]
                L.call("pcrl_pack_conv3_weight", w.detach(), self.fwd, self.dgrad, Co, Ci, dtype_code(dtype), s)
            else:
                Ci, Co = w.shape[0], w.shape[1]
                L.call("pcrl_pack_convt_weight", w.detach(), self.fwd, self.dgrad, Ci, Co, dtype_code(dtype), s)
            self.key = key
            self._built(w.device)
            _record_build(self, (w, dtype))
        else:
            self._reading(w.device)
        return self.fwd, self.dgrad


# ----------------------------------------------------------------------------------------------
# BatchNorm(+activation) on conv outputs
# ----------------------------------------------------------------------------------------------
def bn_eval_coef(gamma, beta, running_mean, running_var):
    """Eval-mode BatchNorm (nn.Module.eval(): running statistics, no update) as the per-channel (scale, shift) the apply kernels take.
    C-element vector arithmetic on the device -- the normalisation itself runs in pcrl_bn_act_apply."""
    scale = (gamma.detach().float() * torch.rsqrt(running_var.float() + BN_EPS)).contiguous()
    shift = (beta.detach().float() - running_mean.float() * scale).contiguous()
    return scale, shift


def bn_finalize(partial, rows, C, count, gamma, beta, running_mean, running_var, training=True):
    if not training:
        scale, shift = bn_eval_coef(gamma, beta, running_mean, running_var)
        return None, None, scale, shift
    dev = partial.device
    if rows > 20000 and C % 2 == 0:
        # full-resolution 2D layers leave > 100 000 statistics rows for as few as 16 channels: pcrl_bn_finalize runs one block per
        # channel, so the rows are first summed by the tiled column-sum kernels (chip-wide, coalesced)
        L = lib()
        both = _f32(2 * C, dev)
        nb = L.call("pcrl_colsum_ws_bytes", rows, 2 * C)
        L.call("pcrl_colsum", partial, both, workspace(nb, dev), nb, rows, 2 * C, dtype_code(torch.float32), stream_handle())
        partial, rows = both, 1
    coef = _f32(4 * C, dev)
    mean, rstd, scale, shift = coef[:C], coef[C:2 * C], coef[2 * C:3 * C], coef[3 * C:]
    order_rmw(running_mean)      # the passes of a step update the running statistics in the reference's order, whatever stream they run on
    lib().call("pcrl_bn_finalize", partial, rows, C, float(count), gamma, beta, running_mean, running_var,
               BN_MOMENTUM, BN_EPS, mean, rstd, scale, shift, stream_handle())
    mark_rmw(running_mean)
    return mean, rstd, scale, shift


def bn_act_apply(y, scale, shift, M, C, act, dtype, out=None):
    a = torch.empty_like(y) if out is None else out
    lib().call("pcrl_bn_act_apply", y, a, scale, shift, M, C, act, dtype_code(dtype), stream_handle())
    return a


def bn_rowadd_ok(C, dtype) -> bool:
    return bool(lib().call("pcrl_bn_act_bwd_rowadd_ok", C, dtype_code(dtype)))


def bn_pool_ok(D, H, W, C, dtype) -> bool:
    return bool(lib().call("pcrl_bn_act_bwd_pool_ok", D, H, W, C, dtype_code(dtype)))


def bn_act_backward(da, y, gamma, mean, rstd, scale, shift, M, C, act, dtype, row_g=None, pool_dp=None, da2=None, pre_partial=None):
    """-> (dy, dgamma, dbeta): gradient w.r.t. the pre-normalisation tensor and the affine parameters.
    `row_g` (float32 [N, C]): the incoming gradient is da + row_g[n] / S broadcast over the S = M / N voxels of a sample (the
    global-average-pool branch, folded into both passes instead of materialised by gap_backward); `da` may then be None.
    `pool_dp` (activation [N, C, D/2, H/2, W/2]): the activation was consumed through MaxPool3d(2) only and THIS is the gradient of
    the pooled tensor (da must be None): max_pool3d_backward happens inside the two passes.
    `da2`: a second gradient tensor of da's shape added to it inside both passes (two consumers of the activation; pcrl_bn_act_bwd_*_sum).
    `pre_partial` = (partial, rows): the first pass was already taken -- by the data-gradient kernel that PRODUCED da, from its output
    tiles (pcrl_conv3d_k3_dgrad_bnred, luconv_backward's `bnred`); plain da only."""
    L, s, dev = lib(), stream_handle(), y.device
    if pre_partial is not None:
        if row_g is not None or pool_dp is not None or da2 is not None:
            raise PcrlError("bn_act_backward: a precomputed first pass covers the plain gradient tensor only")
        partial, rows = pre_partial
    elif pool_dp is not None:
        N, D, H, W, _ = dims(y)
        rows = L.call("pcrl_bn_act_bwd_pool_partial_rows", N, D, H, W)
        partial = _f32(rows * C * 2, dev)
        L.call("pcrl_bn_act_bwd_reduce_pool", pool_dp, y, scale, shift, mean, rstd, partial, N, D, H, W, C, act, dtype_code(dtype), s)
    else:
        rows = L.call("pcrl_bn_bwd_partial_rows", M)
        partial = _f32(rows * C * 2, dev)
        if da2 is not None:
            N = row_g.shape[0] if row_g is not None else 1
            L.call("pcrl_bn_act_bwd_reduce_sum", da, da2, row_g, N, M // N, y, scale, shift, mean, rstd, partial, M, C, act, dtype_code(dtype), s)
        elif row_g is not None:
            N = row_g.shape[0]
            L.call("pcrl_bn_act_bwd_reduce_rowadd", da, row_g, N, M // N, y, scale, shift, mean, rstd, partial, M, C, act, dtype_code(dtype), s)
        else:
            L.call("pcrl_bn_act_bwd_reduce", da, y, scale, shift, mean, rstd, partial, M, C, act, dtype_code(dtype), s)
    out = _f32(5 * C, dev)
    dgamma, dbeta, k1, kB, kA = (out[i * C:(i + 1) * C] for i in range(5))
    L.call("pcrl_bn_bwd_finalize", partial, rows, C, float(M), gamma, mean, rstd, dgamma, dbeta, k1, kB, kA, s)
    dy = torch.empty_like(y)
    if pool_dp is not None:
        L.call("pcrl_bn_act_bwd_apply_pool", pool_dp, y, dy, scale, shift, k1, kB, kA, N, D, H, W, C, act, dtype_code(dtype), s)
    elif da2 is not None:
        L.call("pcrl_bn_act_bwd_apply_sum", da, da2, row_g, N, M // N, y, dy, scale, shift, k1, kB, kA, M, C, act, dtype_code(dtype), s)
    elif row_g is not None:
        L.call("pcrl_bn_act_bwd_apply_rowadd", da, row_g, row_g.shape[0], M // row_g.shape[0], y, dy, scale, shift, k1, kB, kA, M, C, act,
               dtype_code(dtype), s)
    else:
        L.call("pcrl_bn_act_bwd_apply", da, y, dy, scale, shift, k1, kB, kA, M, C, act, dtype_code(dtype), s)
    return dy, dgamma, dbeta


# ----------------------------------------------------------------------------------------------
# LUConv = conv3x3x3 + BatchNorm3d(train) + activation      (models/pcrlv2_model_3d.py:6-34)
# ----------------------------------------------------------------------------------------------
class LUConvSaved:
    __slots__ = ("kind", "x", "y", "mean", "rstd", "scale", "shift", "geom", "act", "gn", "prelu", "z", "dslope", "in_head", "pre_partial", "__weakref__")


def prelu_forward(z, slope, dtype):
    """nn.PReLU(C) on an activation (constructor variant act='prelu', models/pcrlv2_model_3d.py:22-23)."""
    N, D, H, W, C = dims(z)
    a = torch.empty_like(z)
    lib().call("pcrl_prelu_fwd", z, slope.detach(), a, N * D * H * W, C, dtype_code(dtype), stream_handle())
    return a


def prelu_backward(da, z, slope, dtype):
    """-> (dz, dslope [C] float32)."""
    L, s, dev = lib(), stream_handle(), z.device
    N, D, H, W, C = dims(z)
    M = N * D * H * W
    rows = L.call("pcrl_prelu_bwd_partial_rows", M)
    partial = _f32(rows * C, dev)
    dz = torch.empty_like(z)
    L.call("pcrl_prelu_bwd", da, z, slope.detach(), dz, partial, M, C, dtype_code(dtype), s)
    dslope = _f32(C, dev)
    nb = L.call("pcrl_colsum_ws_bytes", rows, C)
    L.call("pcrl_colsum", partial, dslope, workspace(nb, dev), nb, rows, C, dtype_code(torch.float32), s)
    return dz, dslope


_padded_first = {}      # id(conv weight) -> [weakref to it, key, padded float32 copy]


def pad_first_layer(x, conv_w, ci_pad, dtype):
    """in_channels != 1 (constructor variant, models/pcrlv2_model_3d.py:98,102): the input and the first convolution's weight zero-padded
    to `ci_pad` input channels so that the layer runs on the Ci % 32 == 0 implicit-GEMM kernels.  Data movement only.
    The padded weight is ONE persistent tensor per parameter, refreshed IN PLACE when the parameter changed -- keyed on the real parameter
    (`_weights_epoch`, its version counter, its address), so that any way of updating it (FusedSGD's epoch bump, torch.optim's in-place
    update, a manual copy_) is seen; the in-place refresh advances the padded tensor's own version counter, which is what the packed-weight
    cache downstream keys on (a fresh temporary per call would come back at the same address with the same version and hit a stale pack:
    ADVICE r3).  The padded copy is not a parameter and is excluded from the start-of-step prepack plan (it is refreshed here, in forward)."""
    import weakref
    N, Ci, D, H, W = x.shape
    xp = torch.zeros((N, D, H, W, ci_pad), dtype=dtype, device=x.device).permute(0, 4, 1, 2, 3)
    xp[:, :Ci] = x
    key = (_weights_epoch, conv_w._version, conv_w.data_ptr())
    ent = _padded_first.get(id(conv_w))
    if ent is None or ent[0]() is not conv_w or ent[2].shape[1] != ci_pad or ent[2].device != conv_w.device:
        for k in [k for k, e in _padded_first.items() if e[0]() is None]:
            del _padded_first[k]
        wp = torch.zeros((conv_w.shape[0], ci_pad, 3, 3, 3), dtype=torch.float32, device=conv_w.device)
        wp._pcrl_no_prepack = True
        ent = _padded_first[id(conv_w)] = [weakref.ref(conv_w), None, wp]
    if ent[1] != key:
        ent[2][:, :Ci].copy_(conv_w.detach())       # in place: bumps ent[2]._version
        ent[1] = key
    return xp, ent[2]


def luconv_forward(x, conv_w, conv_b, gamma, beta, running_mean, running_var, packed: PackedWeights, act: int, dtype, training=True, gn_groups=0,
                   pooled=False, gap=False, prelu=None, inorm=False, pool_only=False):
    """x: activation (or float32 [N,1,D,H,W] for the first layer).  Returns (a, saved); with `pooled` ((a, MaxPool3d(2)(a)), saved) --
    one pass where bn_pool_ok (BatchNorm layers of the MFMA path), else the separate pool; with `gap` ((a, global average pool [N, C]
    float32 of a), saved), likewise in one pass where bn_rowadd_ok.
    training=False: eval mode -- the normalisation uses the running statistics, nothing is updated, no statistics are gathered."""
    L, s, dev = lib(), stream_handle(), x.device
    Co, Ci = conv_w.shape[0], conv_w.shape[1]
    sv = LUConvSaved()
    # prelu (a float32 [Co] slope vector; constructor variant act='prelu'): the normalisation runs without an activation and
    # pcrl_prelu_fwd follows as its own pass -- none of the fused apply kernels applies
    if prelu is not None:
        act = ACT_NONE
    sv.act = act
    sv.gn = None
    sv.prelu, sv.z, sv.dslope, sv.in_head = prelu, None, None, False
    if inorm and Co == 1:
        # norm='in' deep-supervision head (InstanceNorm3d(1), models/pcrlv2_model_3d.py:15-16,60): per-sample statistics of a 1-channel
        # float32 map -- the GroupNorm machinery on the map viewed as [N][S / 4][4] with ONE group over its four pseudo-channels
        N, D, H, W, C = dims(x)
        M, S = N * D * H * W, D * H * W
        if S % 4:
            raise PcrlError("InstanceNorm head: D*H*W must be a multiple of 4")
        y = _f32(M, dev)
        nb = L.call("pcrl_conv3d_to1_fwd_ws_bytes", N, D, H, W, C, 27)
        L.call("pcrl_conv3d_to1_fwd", x, conv_w.detach(), conv_b.detach(), y, None, workspace(nb, dev) if nb else None, nb,
               N, D, H, W, C, 27, dtype_code(dtype), s)
        g4, b4 = gamma.detach().expand(4).contiguous(), beta.detach().expand(4).contiguous()
        a, saved = gn_forward_rows(y.view(N, S // 4, 4), g4, b4, 1, act, torch.float32)
        sv.gn, sv.in_head, sv.kind = (saved, 1), True, "to1"
        sv.x, sv.y, sv.mean, sv.rstd, sv.scale, sv.shift = x, y, None, None, None, None
        sv.geom = (N, D, H, W, Ci, Co)
        return a.view(N, 1, D, H, W), sv
    if gn_groups and Co > 1:
        # OPTIONAL, NOT IN THE REFERENCE (north_star's GroupNorm + SiLU; the reference's own norm='gn' crashes):
        # conv -> GroupNorm(groups) -> activation.  Per-sample statistics: the same in train and eval mode, no running buffers.
        N, D, H, W = (x.shape[0], x.shape[2], x.shape[3], x.shape[4])
        y = new_act(N, D, H, W, Co, dtype, dev)
        if Ci == 1:
            L.call("pcrl_conv3d_k3_c1_fwd", x, conv_w.detach(), conv_b.detach(), y, None, N, D, H, W, Co, dtype_code(dtype), s)
            sv.kind = "c1"
        else:
            wf, _ = packed.get(conv_w, dtype)
            nb = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dtype))
            L.call("pcrl_conv3d_k3_fwd_ws", x, wf, conv_b.detach(), y, None, workspace(nb, dev) if nb else None, nb, N, D, H, W, Ci, Co,
                   dtype_code(dtype), s)
            sv.kind = "gemm"
        a, saved = gn_act_forward(y, gamma.detach(), beta.detach(), gn_groups, act, dtype)
        sv.gn = (saved, gn_groups)
        sv.x, sv.y, sv.mean, sv.rstd, sv.scale, sv.shift = x, y, None, None, None, None
        sv.geom = (N, D, H, W, Ci, Co)
        if prelu is not None:
            sv.z, a = a, prelu_forward(a, prelu, dtype)
        if pooled:
            a = (a, maxpool_forward(a, dtype))
        elif gap:
            a = (a, gap_forward(a, dtype))
        return a, sv
    if Co == 1:  # deep-supervision head: C -> 1, float32 map out
        N, D, H, W, C = dims(x)
        if C != Ci:
            raise PcrlError(f"LUConv: input has {C} channels, weight expects {Ci}")
        M = N * D * H * W
        rows = L.call("pcrl_conv3d_to1_stats_rows", N, D, H, W, C, 27, dtype_code(dtype))
        y = _f32(M, dev)
        partial = _f32(rows * 2, dev) if training else None
        nb = L.call("pcrl_conv3d_to1_fwd_ws_bytes", N, D, H, W, C, 27)
        L.call("pcrl_conv3d_to1_fwd", x, conv_w.detach(), conv_b.detach(), y, partial, workspace(nb, dev) if nb else None, nb,
               N, D, H, W, C, 27, dtype_code(dtype), s)
        mean, rstd, scale, shift = bn_finalize(partial, rows, 1, M, gamma.detach(), beta.detach(), running_mean, running_var, training)
        a = bn_act_apply(y, scale, shift, M, 1, act, torch.float32).view(N, 1, D, H, W)
        sv.kind = "to1"
    elif Ci == 1:  # first layer: 1 -> Co on a float32 scalar field
        if x.dtype != torch.float32 or not x.is_contiguous() or x.shape[1] != 1:
            raise PcrlError("first-layer input must be a contiguous float32 [N,1,D,H,W] tensor")
        N, _, D, H, W = x.shape
        M = N * D * H * W
        rows = L.call("pcrl_conv3d_k3_c1_stats_rows", N, D, H, W, Co, dtype_code(dtype))
        y = new_act(N, D, H, W, Co, dtype, dev)
        partial = _f32(rows * Co * 2, dev) if training else None
        L.call("pcrl_conv3d_k3_c1_fwd", x, conv_w.detach(), conv_b.detach(), y, partial, N, D, H, W, Co, dtype_code(dtype), s)
        mean, rstd, scale, shift = bn_finalize(partial, rows, Co, M, gamma.detach(), beta.detach(), running_mean, running_var, training)
        a = bn_act_apply(y, scale, shift, M, Co, act, dtype)
        sv.kind = "c1"
    else:
        N, D, H, W, C = dims(x)
        if C != Ci:
            raise PcrlError(f"LUConv: input has {C} channels, weight expects {Ci}")
        if x.dtype != dtype:
            raise PcrlError(f"LUConv: activation dtype {x.dtype} != compute dtype {dtype}")
        M = N * D * H * W
        rows = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dtype))
        wf, _ = packed.get(conv_w, dtype)
        y = new_act(N, D, H, W, Co, dtype, dev)
        partial = _f32(rows * Co * 2, dev) if training else None
        nb = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dtype))
        L.call("pcrl_conv3d_k3_fwd_ws", x, wf, conv_b.detach(), y, partial, workspace(nb, dev) if nb else None, nb, N, D, H, W, Ci, Co,
               dtype_code(dtype), s)
        mean, rstd, scale, shift = bn_finalize(partial, rows, Co, M, gamma.detach(), beta.detach(), running_mean, running_var, training)
        if pooled and prelu is None and config.FUSE_APPLY_CONSUMERS and bn_pool_ok(D, H, W, Co, dtype):
            # pool_only (the engine's training step): the unpooled activation is not stored at all -- nothing reads it (the backward of this pair
            # reads y; PCRLv23d hands the stashed attribute out lazily, from y, if somebody asks)
            a, p = None if pool_only else torch.empty_like(y), new_act(N, D // 2, H // 2, W // 2, Co, dtype, dev)
            L.call("pcrl_bn_act_apply_pool", y, a, p, scale, shift, N, D, H, W, Co, act, dtype_code(dtype), s)
            a, pooled = (a, p), False
        elif gap and prelu is None and config.FUSE_APPLY_CONSUMERS and bn_rowadd_ok(Co, dtype):
            a, g = torch.empty_like(y), _f32(N * Co, dev).view(N, Co)
            nbg = L.call("pcrl_gap_ws_bytes", N, D * H * W, Co)
            L.call("pcrl_bn_act_apply_gap", y, a, g, scale, shift, workspace(nbg, dev), nbg, N, D * H * W, Co, act, dtype_code(dtype), s)
            a, gap = (a, g), False
        else:
            a = bn_act_apply(y, scale, shift, M, Co, act, dtype)
        sv.kind = "gemm"
    sv.x, sv.y, sv.mean, sv.rstd, sv.scale, sv.shift = x, y, mean, rstd, scale, shift
    sv.geom = (N, D, H, W, Ci, Co)
    if prelu is not None:
        sv.z, a = a, prelu_forward(a, prelu, dtype)
    if pooled:
        a = (a, maxpool_forward(a, dtype))
    elif gap:
        a = (a, gap_forward(a, dtype))
    return a, sv


def take_pre_partial(sv: LUConvSaved, da):
    """The first BatchNorm-backward pass of `sv`'s layer if the data gradient above it already took it for exactly this gradient tensor
    (luconv_backward's `bnred`), else None.  One-shot."""
    pre = getattr(sv, "pre_partial", None)
    sv.pre_partial = None
    if pre is None or da is None or pre[2].data_ptr() != da.data_ptr() or pre[2].shape != da.shape:
        return None
    return pre[0], pre[1]


def luconv_backward(sv: LUConvSaved, da, conv_w, gamma, packed: PackedWeights, dtype, need_dx=True, dx_add=None, dx_colsum=None, da_row_g=None,
                    pool_dp=None, bnred: LUConvSaved = None):
    """-> (dx | None, dw, db, dgamma, dbeta).  `dx_add`: optional activation folded into dx (to1 kind only).
    `da_row_g`: optional float32 [N, Co] -- the gradient of this LUConv's output is da + da_row_g[n] / (D*H*W) (bn_act_backward; da may
    be None then).  Only for BatchNorm layers with bn_rowadd_ok(Co, dtype).
    `pool_dp`: the output was consumed through MaxPool3d(2) only; this is the pooled tensor's gradient and da is None
    (bn_act_backward; BatchNorm layers with bn_pool_ok).
    `dx_colsum`: optional float32 [Ci] that receives sum over voxels of dx (the bias gradient of the layer that produced this
    LUConv's input -- ConvTranspose3d in UpTransition), taken from the data-gradient kernel's float accumulators through its
    per-tile statistics output instead of re-reading dx from HBM.

    db is exactly zero: a bias that is followed by a batch-statistics normalisation has an identically
    zero gradient (the reference's autograd yields round-off noise ~1e-8 there).

    `bnred`: the saved state of the LUConv BELOW whose activation is this convolution's input and has NO other consumer (ops.0 under ops.1
    in nn.Sequential(LUConv, LUConv), models/pcrlv2_model_3d.py:37-45): where the library has the kernel
    (pcrl_conv3d_k3_dgrad_bnred_rows) the data gradient takes the first pass of that layer's BatchNorm backward from its output tiles
    and leaves it on `bnred.pre_partial` for that layer's backward (take_pre_partial); otherwise nothing changes.
    """
    L, s = lib(), stream_handle()
    N, D, H, W, Ci, Co = sv.geom
    M = N * D * H * W
    dev = sv.y.device
    dw = torch.empty_like(conv_w, dtype=torch.float32, memory_format=torch.contiguous_format)
    db = zero_grad_vector(Co, dev)
    if sv.prelu is not None:      # act='prelu': through the PReLU pass first (the callers fold nothing into `da` in this mode)
        if da is None or da_row_g is not None or pool_dp is not None:
            raise PcrlError("luconv_backward: the PReLU variant takes the complete gradient of the activation as a tensor")
        da, sv.dslope = prelu_backward(da, sv.z, sv.prelu, dtype)
    if sv.kind == "to1":
        da = da.contiguous()
        if sv.in_head:
            S = D * H * W
            dy, dg4, db4 = gn_backward_rows(da.view(N, S // 4, 4), sv.gn[0], gamma.detach().expand(4).contiguous(), 1, sv.act, torch.float32)
            dy, dgamma, dbeta = dy.view(-1), dg4.sum().view(1), db4.sum().view(1)
        else:
            dy, dgamma, dbeta = bn_act_backward(da, sv.y, gamma.detach(), sv.mean, sv.rstd, sv.scale, sv.shift, M, 1, sv.act, torch.float32)
        nb = L.call("pcrl_conv3d_to1_wgrad_ws_bytes", N, D, H, W, Ci, 27)
        dbias_unused = _f32(1, dev)
        # (inline on the data-gradient chain; on the side stream like the other weight gradients it measured 31.80 vs 31.70 ms: not kept)
        L.call("pcrl_conv3d_to1_wgrad", sv.x, dy, dw, dbias_unused, workspace(nb, dev), nb, N, D, H, W, Ci, 27, dtype_code(dtype), s)
        dx = None
        if need_dx:
            dx = new_act(N, D, H, W, Ci, dtype, dev)
            L.call("pcrl_conv3d_to1_dgrad", dy, conv_w.detach(), dx_add, dx, N, D, H, W, Ci, 27, dtype_code(dtype), s)
        return dx, dw, db, dgamma, dbeta
    if da is not None:
        dims(da)
    if sv.gn is not None:   # optional GroupNorm mode: the conv bias is NOT cancelled by the normalisation -> db = column sums of dy
        dy, dgamma, dbeta = gn_act_backward(da, sv.gn[0], gamma.detach(), sv.gn[1], sv.act, dtype)
        db = _f32(Co, dev)
        nbc = L.call("pcrl_colsum_ws_bytes", M, Co)
        L.call("pcrl_colsum", dy, db, workspace(nbc, dev), nbc, M, Co, dtype_code(dtype), s)
    else:
        pre = take_pre_partial(sv, da) if (da_row_g is None and pool_dp is None) else None
        dy, dgamma, dbeta = bn_act_backward(da, sv.y, gamma.detach(), sv.mean, sv.rstd, sv.scale, sv.shift, M, Co, sv.act, dtype,
                                            row_g=da_row_g, pool_dp=pool_dp, pre_partial=pre)
    if sv.kind == "c1":
        nb = L.call("pcrl_conv3d_k3_c1_wgrad_ws_bytes", N, D, H, W, Co)
        L.call("pcrl_conv3d_k3_c1_wgrad", sv.x, dy, dw, workspace(nb, dev), nb, N, D, H, W, Co, dtype_code(dtype), s)
        return None, dw, db, dgamma, dbeta
    def weight_gradient():
        nbw = L.call("pcrl_conv3d_k3_wgrad_ws_bytes", N, D, H, W, Ci, Co)
        with side_wgrad(dev, sv.x, dy) as ws:
            L.call("pcrl_conv3d_k3_wgrad", sv.x, dy, dw, ws(nbw), nbw, N, D, H, W, Ci, Co, dtype_code(dtype), stream_handle())

    # queued IN FRONT of the data gradient: both need dy, the two matrix kernels then share the chip (queued behind it -- next to the BatchNorm
    # backward of the layer below -- was an experiment of round 3, bit-identical, no gain: removed)
    weight_gradient()
    dx = None
    if need_dx:
        _, wd = packed.get(conv_w, dtype)
        dx = new_act(N, D, H, W, Ci, dtype, dev)
        nb = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Co, Ci, dtype_code(dtype))
        part = None
        if dx_colsum is not None:
            rows = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Co, Ci, dtype_code(dtype))
            part = _f32(rows * Ci * 2, dev)
        brows = 0
        if (bnred is not None and part is None and config.DGRAD_BNRED and bnred.gn is None and getattr(bnred, "prelu", None) is None and bnred.mean is not None
                and bnred.y.dtype == dtype and bnred.y.numel() == M * Ci):
            brows = L.call("pcrl_conv3d_k3_dgrad_bnred_rows", N, D, H, W, Co, Ci, bnred.act, dtype_code(dtype))
        if brows:
            bpart = _f32(brows * Ci * 2, dev)
            L.call("pcrl_conv3d_k3_dgrad_bnred", dy, wd, dx, bnred.y, bnred.scale, bnred.shift, bnred.mean, bnred.rstd, bpart,
                   N, D, H, W, Co, Ci, bnred.act, dtype_code(dtype), s)
            bnred.pre_partial = (bpart, brows, dx)
        else:
            L.call("pcrl_conv3d_k3_fwd_ws", dy, wd, None, dx, part, workspace(nb, dev) if nb else None, nb, N, D, H, W, Co, Ci, dtype_code(dtype), s)
        if part is not None:   # [rows][Ci][2] -> column sums; the (sum) entries are the even columns
            both = _f32(Ci * 2, dev)
            nb2 = L.call("pcrl_colsum_ws_bytes", rows, Ci * 2)
            L.call("pcrl_colsum", part, both, workspace(nb2, dev), nb2, rows, Ci * 2, dtype_code(torch.float32), s)
            dx_colsum.copy_(both.view(Ci, 2)[:, 0])
    return dx, dw, db, dgamma, dbeta


# ----------------------------------------------------------------------------------------------
# MaxPool3d(2), ConvTranspose3d(k2,s2), global average pool
# ----------------------------------------------------------------------------------------------
def maxpool_forward(x, dtype):
    N, D, H, W, C = dims(x)
    y = new_act(N, D // 2, H // 2, W // 2, C, dtype, x.device)
    lib().call("pcrl_maxpool3d_2_fwd", x, y, N, D, H, W, C, dtype_code(dtype), stream_handle())
    return y


def maxpool_backward(x, dy, dtype):
    N, D, H, W, C = dims(x)
    dims(dy)
    dx = torch.empty_like(x)
    lib().call("pcrl_maxpool3d_2_bwd", x, dy, dx, N, D, H, W, C, dtype_code(dtype), stream_handle())
    return dx


def convt_forward(x, w, b, packed: PackedWeights, dtype):
    N, D, H, W, Ci = dims(x)
    Co = w.shape[1]
    wf, _ = packed.get(w, dtype)
    y = new_act(N, 2 * D, 2 * H, 2 * W, Co, dtype, x.device)
    lib().call("pcrl_convt3d_k2s2_fwd", x, wf, b.detach(), y, N, D, H, W, Ci, Co, dtype_code(dtype), stream_handle())
    return y


def convt_backward(x, dy, w, packed: PackedWeights, dtype, need_dx=True, db=None):
    """-> (dx | None, dw [Ci,Co,2,2,2], db [Co]).  `db`: bias gradient if the caller already has it (luconv_backward's dx_colsum)."""
    L, s, dev = lib(), stream_handle(), x.device
    N, D, H, W, Ci = dims(x)
    Co = w.shape[1]
    dims(dy)
    dw = torch.empty_like(w, dtype=torch.float32, memory_format=torch.contiguous_format)
    nb = L.call("pcrl_convt3d_k2s2_wgrad_ws_bytes", N, D, H, W, Ci, Co)
    with side_wgrad(dev, x, dy) as ws:
        L.call("pcrl_convt3d_k2s2_wgrad", x, dy, dw, ws(nb), nb, N, D, H, W, Ci, Co, dtype_code(dtype), stream_handle())
    if db is None:
        Mo = N * D * H * W * 8
        db = _f32(Co, dev)
        nb2 = L.call("pcrl_colsum_ws_bytes", Mo, Co)
        L.call("pcrl_colsum", dy, db, workspace(nb2, dev), nb2, Mo, Co, dtype_code(dtype), s)
    dx = None
    if need_dx:
        _, wd = packed.get(w, dtype)
        dx = new_act(N, D, H, W, Ci, dtype, dev)
        L.call("pcrl_convt3d_k2s2_dgrad", dy, wd, dx, N, D, H, W, Ci, Co, dtype_code(dtype), s)
    return dx, dw, db


# ----------------------------------------------------------------------------------------------
# UpTransition: ConvTranspose3d(k2,s2) -> conv1 of ops.0 as one operator on the coarse grid   (models/pcrlv2_model_3d.py:64; csrc/upconv_fused.hip)
# ----------------------------------------------------------------------------------------------
class ComposedUpConv(_CacheGuard):
    """Composed weights of (up_conv, ops.0.conv1), rebuilt when either parameter changed (once per optimizer step), and the
    accumulators of their gradients: every backward pass adds its gradient of the COMPOSED weights (and the border-class sums of dy0)
    here; the chain rule to the two reference parameters is linear in those, so it runs once per backward() call, from the engine's
    end-of-backward callback (deliver_composed), however many forward passes shared the weights."""

    def __init__(self):
        self.key = None
        self.wf = self.wd = self.w3f = self.wd3 = self.bias_tab = None
        self.want = (False, False)      # sticky: some pass of the step runs forward / data gradient on the wide-brick kernel (3x3x3 weight forms)
        self.dweff = self.box = None
        self.pending = None     # (w_up, b_up, w0, dtype) while accumulated gradients wait for delivery

    def get(self, w_up, b_up, w0, b0, dtype, geom=None):
        """-> (wf, wd, bias_tab); .w3f / .wd3 hold the 3x3x3 forms for the wide-brick kernels.  geom = (N, D, H, W) of a forward pass: the
        3x3x3 forms are produced from then on if that shape runs on the brick kernels (passes of other shapes share the same weights)."""
        if geom is not None:
            Ci, Co, L = w_up.shape[0], w0.shape[0], lib()
            f = bool(L.call("pcrl_upconv_fwd_uses_brick", *geom, Ci, Co, dtype_code(dtype)))
            d = bool(L.call("pcrl_upconv_dgrad_uses_brick", *geom, Ci, Co, dtype_code(dtype)))
            self.want = (self.want[0] or f, self.want[1] or d)
        key = (_weights_epoch, w_up._version, w_up.data_ptr(), b_up._version, w0._version, w0.data_ptr(), b0._version, dtype, self.want)
        if key != self.key:
            L, s, dev = lib(), stream_handle(), w_up.device
            Ci, Cm, Co = w_up.shape[0], w_up.shape[1], w0.shape[0]
            if self.key is not None and key[:-1] == self.key[:-1]:
                # same weights, a later pass of the step (another shape) needs more weight forms: a pass on another stream may still be READING
                # the tensors this rebuild drops -- keep the allocator from recycling them under it
                for t in (self.wf, self.wd, self.w3f, self.wd3, self.bias_tab):
                    if t is not None:
                        for st in list(_side_streams.values()) + list(_view_streams.values()):
                            t.record_stream(st)
            if w0.shape[1] != Cm:
                raise PcrlError(f"composed up-conv: up_conv has {Cm} output channels, conv1 expects {w0.shape[1]}")
            self.wf = torch.empty(64 * Ci * Co, dtype=dtype, device=dev)
            self.wd = torch.empty(64 * Ci * Co, dtype=dtype, device=dev)
            self.w3f = torch.empty(216 * Ci * Co, dtype=dtype, device=dev) if self.want[0] else None
            self.wd3 = torch.empty(216 * Ci * Co, dtype=dtype, device=dev) if self.want[1] else None
            self.bias_tab = _f32(27 * Co, dev)
            nb = L.call("pcrl_upconv_compose_ws_bytes", Ci, Cm, Co, dtype_code(dtype))
            L.call("pcrl_upconv_compose", w_up.detach(), b_up.detach(), w0.detach(), b0.detach(), self.wf, self.wd, self.w3f, self.wd3, self.bias_tab,
                   workspace(nb, dev), nb, Ci, Cm, Co, dtype_code(dtype), s)
            self.key = key
            self._built(dev)
            _record_build(self, (w_up, b_up, w0, b0, dtype))
        else:
            self._reading(w_up.device)
        return self.wf, self.wd, self.bias_tab

    def accumulate(self, x, dy, geom, w_up, b_up, w0, dtype, zero_sum=False):
        """One backward pass: dweff += d(composed weights), box += border-class sums of dy (pcrl_upconv_wgrad_accum).  zero_sum: dy is the
        output of a training-mode BatchNorm backward over exactly these voxels (its per-channel sum is zero in exact arithmetic)."""
        L, dev = lib(), x.device
        N, D, H, W, Ci, Co = geom
        first = self.pending is None
        if first:
            if self.dweff is None or self.dweff.numel() != 64 * Ci * Co or self.dweff.device != dev:
                self.dweff, self.box = _f32(64 * Ci * Co, dev), _f32(27 * Co, dev)
            self.pending = (w_up, b_up, w0, dtype)
            _pending_composed.append(self)
        nb = L.call("pcrl_upconv_wgrad_accum_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dtype))
        with side_wgrad(dev, x, dy, shared_accumulator=True) as ws:
            L.call("pcrl_upconv_wgrad_accum", x, dy, self.dweff, self.box, (1 if first else 0) | (2 if zero_sum else 0), ws(nb), nb, N, D, H, W, Ci, Co,
                   dtype_code(dtype), stream_handle())

    def finish(self):
        """-> (w_up, b_up, w0, dw_up, db_up, dw0) for everything accumulated since the last delivery."""
        w_up, b_up, w0, dtype = self.pending
        self.pending = None
        L, dev = lib(), w_up.device
        Ci, Cm, Co = w_up.shape[0], w_up.shape[1], w0.shape[0]
        dw_up = torch.empty_like(w_up, dtype=torch.float32, memory_format=torch.contiguous_format)
        dw0 = torch.empty_like(w0, dtype=torch.float32, memory_format=torch.contiguous_format)
        db_up = _f32(Cm, dev)
        nb = L.call("pcrl_upconv_wgrad_finish_ws_bytes", Ci, Cm, Co, dtype_code(dtype))
        L.call("pcrl_upconv_wgrad_finish", self.dweff, self.box, w_up.detach(), b_up.detach(), w0.detach(), dw_up, db_up, dw0, workspace(nb, dev), nb,
               Ci, Cm, Co, dtype_code(dtype), stream_handle())
        return w_up, b_up, w0, dw_up, db_up, dw0


_pending_composed: list = []


def deliver_composed(only=None):
    """End of a backward() call (or, `only` = one ComposedUpConv, the moment its last pass has accumulated): run the chain rule of every
    composed up-conv that accumulated gradients.  -> [(parameter, gradient), ...] for up_conv.weight, up_conv.bias and ops.0.conv1.weight.
    With the weight-gradient side stream on, the chain rule is queued THERE, behind the accumulations it consumes (same stream: no join) and
    next to whatever the other streams still have to do -- at the end of a backward() it would otherwise be a serial tail on an idle chip;
    its results are parked like every side-stream gradient (the join before they are summed covers them)."""
    out = []
    todo = [c for c in _pending_composed if only is None or c is only]
    if todo:
        dev = None
        for c in todo:
            if c.pending is not None:
                dev = c.pending[0].device
        side = config.EARLY_COMPOSED and dev is not None and side_wgrad(dev).active
        if not side:
            join_side_stream()        # one-stream mode: nothing to wait for, kept for the 2D / CPU-less paths' symmetry
        for c in todo:
            if c.pending is not None:
                if side:
                    with side_wgrad(dev):
                        w_up, b_up, w0, dw_up, db_up, dw0 = c.finish()
                else:
                    w_up, b_up, w0, dw_up, db_up, dw0 = c.finish()
                out += [(w_up, dw_up), (b_up, db_up), (w0, dw0)]
            _pending_composed.remove(c)
    return out


def drop_pending_composed():
    """Start of a step: forget accumulations of a backward pass that raised before its end-of-backward callback ran."""
    for c in _pending_composed:
        c.pending = None
    _pending_composed.clear()


def upconv_luconv_forward(x, w_up, b_up, conv_w, conv_b, gamma, beta, running_mean, running_var, composed: ComposedUpConv, act, dtype):
    """act(bn(conv1(up_conv(x)))) with the two convolutions composed.  x: coarse activation [N, Ci, D, H, W] -> (a [N, Co, 2D, 2H, 2W], saved).
    Training mode, BatchNorm only (the GroupNorm / eval routes keep the two separate kernels)."""
    L, s, dev = lib(), stream_handle(), x.device
    N, D, H, W, Ci = dims(x)
    Co = conv_w.shape[0]
    if w_up.shape[0] != Ci:
        raise PcrlError(f"up_conv: input has {Ci} channels, weight expects {w_up.shape[0]}")
    wf, _, bias_tab = composed.get(w_up, b_up, conv_w, conv_b, dtype, geom=(N, D, H, W))
    M = N * D * H * W * 8
    rows = L.call("pcrl_upconv_stats_rows", N, D, H, W, Ci, Co, dtype_code(dtype))
    y = new_act(N, 2 * D, 2 * H, 2 * W, Co, dtype, dev)
    partial = _f32(rows * Co * 2, dev)
    L.call("pcrl_upconv_fwd", x, wf, composed.w3f, bias_tab, y, partial, N, D, H, W, Ci, Co, dtype_code(dtype), s)
    mean, rstd, scale, shift = bn_finalize(partial, rows, Co, M, gamma.detach(), beta.detach(), running_mean, running_var, True)
    a = bn_act_apply(y, scale, shift, M, Co, act, dtype)
    sv = LUConvSaved()
    sv.kind, sv.act, sv.gn = "upc", act, None
    sv.x, sv.y, sv.mean, sv.rstd, sv.scale, sv.shift = x, y, mean, rstd, scale, shift
    sv.geom = (N, D, H, W, Ci, Co)      # COARSE dims
    return a, sv


def upconv_luconv_backward(sv: LUConvSaved, da, w_up, b_up, conv_w, conv_b, gamma, composed: ComposedUpConv, dtype, need_dx=True, defer=False):
    """-> (dx | None, dw_up, db_up, dw0, db0 (exactly zero: a bias in front of batch statistics), dgamma, dbeta).
    defer=True (the autograd engine's route): the three convolution-parameter gradients come back as None -- this pass's share is added to
    `composed`'s accumulators and delivered once per backward() by deliver_composed()."""
    L, s, dev = lib(), stream_handle(), sv.y.device
    N, D, H, W, Ci, Co = sv.geom
    M = N * D * H * W * 8
    dy, dgamma, dbeta = bn_act_backward(da, sv.y, gamma.detach(), sv.mean, sv.rstd, sv.scale, sv.shift, M, Co, sv.act, dtype,
                                        pre_partial=take_pre_partial(sv, da))
    dx = None

    def data_gradient():
        _, wd, _ = composed.get(w_up, b_up, conv_w, conv_b, dtype)
        dxx = new_act(N, D, H, W, Ci, dtype, dev)
        nbd = L.call("pcrl_upconv_dgrad_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dtype))      # split-K scratch on the small coarse grids
        L.call("pcrl_upconv_dgrad_ws", dy, wd, composed.wd3, dxx, workspace(nbd, dev) if nbd else None, nbd, N, D, H, W, Ci, Co, dtype_code(dtype), s)
        return dxx

    composed.accumulate(sv.x, dy, sv.geom, w_up, b_up, conv_w, dtype, zero_sum=config.UPC_ZERO_SUM)   # dy: output of the BatchNorm backward just above
    dw_up = db_up = dw0 = None
    if not defer:
        join_side_stream()
        _pending_composed.remove(composed)
        _, _, _, dw_up, db_up, dw0 = composed.finish()
    if need_dx:
        dx = data_gradient()
    return dx, dw_up, db_up, dw0, zero_grad_vector(Co, dev), dgamma, dbeta


def gap_forward(a, dtype):
    N, D, H, W, C = dims(a)
    S = D * H * W
    g = _f32(N * C, a.device).view(N, C)
    L = lib()
    nb = L.call("pcrl_gap_ws_bytes", N, S, C)
    L.call("pcrl_gap_fwd", a, g, workspace(nb, a.device), nb, N, S, C, dtype_code(dtype), stream_handle())
    return g


def gap_backward(dg, like, add_src, dtype):
    """da = add_src + dg/S broadcast over the volume (add_src may be None)."""
    N, D, H, W, C = dims(like)
    da = torch.empty_like(like)
    lib().call("pcrl_gap_bwd", dg.contiguous(), add_src, da, N, D * H * W, C, dtype_code(dtype), stream_handle())
    return da


# ----------------------------------------------------------------------------------------------
# heads: BatchNorm1d / Linear on [rows, C] float32
# ----------------------------------------------------------------------------------------------
def bn1d_eval(x, gamma, beta, running_mean, running_var, relu: bool):
    """Eval-mode BatchNorm1d (+ReLU) on [rows, C] float32: the per-channel apply kernel with coefficients from the running statistics."""
    rows, C = x.shape
    scale, shift = bn_eval_coef(gamma, beta, running_mean, running_var)
    return bn_act_apply(x.contiguous(), scale, shift, rows, C, ACT_RELU if relu else ACT_NONE, torch.float32)


def bn1d_forward(x, gamma, beta, running_mean, running_var, relu: bool):
    rows, C = x.shape
    y = torch.empty_like(x)
    st = _f32(2 * C, x.device)
    order_rmw(running_mean)
    lib().call("pcrl_bn1d_fwd", x, y, gamma.detach(), beta.detach(), running_mean, running_var, BN_MOMENTUM, BN_EPS,
               st[:C], st[C:], rows, C, int(relu), stream_handle())
    mark_rmw(running_mean)
    return y, st[:C], st[C:]


def bn1d_backward(dy, x, y, gamma, mean, rstd, relu: bool):
    rows, C = x.shape
    dx = torch.empty_like(x)
    g = _f32(2 * C, x.device)
    lib().call("pcrl_bn1d_bwd", dy.contiguous(), x, y, gamma.detach(), mean, rstd, dx, g[:C], g[C:], rows, C, int(relu), stream_handle())
    return dx, g[:C], g[C:]


def linear_forward(x, w, b):
    rows, Cin = x.shape
    Cout = w.shape[0]
    y = _f32(rows * Cout, x.device).view(rows, Cout)
    lib().call("pcrl_linear_fwd", x, w.detach(), b.detach(), y, rows, Cin, Cout, stream_handle())
    return y


FUSED_HEAD_BWD = True     # A/B switch: 0 = nine launches (Linear backward x 2, BatchNorm1d backward x 2, add)


def heads_backward(d_pro, d_pre, heads, x_pro, params):
    """Backward of x_pro = bn(pooled); x_pre = predictor_head(x_pro) (pcrlv2_model_3d.py:55-59,67-70; pcrlv2_model.py:108-111,124-127) down to
    the pooled vector.  heads = (pooled g, mean / rstd of bn, h0, h1, mean / rstd of predictor_head[1]); params = (bn.weight, bn.bias, ph0.weight,
    ph0.bias, ph1.weight, ph1.bias, ph3.weight, ph3.bias).  -> (d_g float32 [N, C], [their 8 gradients]).
    Two launches (pcrl_head_bwd_stage: Linear backward + the BatchNorm1d backward of what it produced, csrc/heads_fused.hip) where the rows fit
    (N <= 512); else -- or with ops.FUSED_HEAD_BWD = False (the test's handle) -- the separate kernels."""
    bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b = params
    g, m_pro, r_pro, h0, h1, m_h, r_h = heads
    N, C = g.shape
    grads = [None] * 8
    if FUSED_HEAD_BWD and N <= 512 and C % 4 == 0:
        L, s, dev = lib(), stream_handle(), g.device
        H = h0.shape[1]
        d_h0 = None
        if d_pre is not None:
            d_pre = d_pre.contiguous()
            d_h0 = _f32(N * H, dev).view(N, H)
            v = _f32(2 * H + C, dev)
            g_p3w = torch.empty_like(p3_w, dtype=torch.float32)
            L.call("pcrl_head_bwd_stage", d_pre, p3_w.detach(), None, h1, h0, h1, p1_g.detach(), m_h, r_h, d_h0, v[:H], v[H:2 * H], g_p3w, v[2 * H:], N, C, H, 1, s)
            grads[4], grads[5], grads[6], grads[7] = v[:H], v[H:2 * H], g_p3w, v[2 * H:]
        d_g = _f32(N * C, dev).view(N, C)
        v2 = _f32(2 * C + H, dev)
        g_p0w = torch.empty_like(p0_w, dtype=torch.float32) if d_h0 is not None else None
        L.call("pcrl_head_bwd_stage", d_h0, p0_w.detach() if d_h0 is not None else None, d_pro.contiguous() if d_pro is not None else None,
               x_pro if d_h0 is not None else None, g, None, bn_g.detach(), m_pro, r_pro, d_g, v2[:C], v2[C:2 * C], g_p0w,
               v2[2 * C:] if d_h0 is not None else None, N, H, C, 0, s)
        grads[0], grads[1] = v2[:C], v2[C:2 * C]
        if d_h0 is not None:
            grads[2], grads[3] = g_p0w, v2[2 * C:]
        return d_g, grads
    d_xpro = d_pro.contiguous() if d_pro is not None else None
    if d_pre is not None:
        d_h1, g_p3w, g_p3b = linear_backward(d_pre, h1, p3_w)
        d_h0, g_p1g, g_p1b = bn1d_backward(d_h1, h0, h1, p1_g, m_h, r_h, relu=True)
        d_xp, g_p0w, g_p0b = linear_backward(d_h0, x_pro, p0_w)
        d_xpro = d_xp if d_xpro is None else add2_small(d_xpro, d_xp)
        grads[2:8] = [g_p0w, g_p0b, g_p1g, g_p1b, g_p3w, g_p3b]
    d_g, g_bng, g_bnb = bn1d_backward(d_xpro, g, x_pro, bn_g, m_pro, r_pro, relu=False)
    grads[0], grads[1] = g_bng, g_bnb
    return d_g, grads


def add2_small(a, b):
    """a + b for two float32 tensors of one shape (head-sized matrices), on the library's kernel instead of aten::add."""
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    lib().call("pcrl_add_f32", a, b, out, a.numel(), stream_handle())
    return out


def linear_backward(dy, x, w):
    rows, Cin = x.shape
    Cout = w.shape[0]
    dx = torch.empty_like(x)
    dw = torch.empty_like(w, dtype=torch.float32)
    db = _f32(Cout, x.device)
    lib().call("pcrl_linear_bwd", dy.contiguous(), x, w.detach(), dx, dw, db, rows, Cin, Cout, stream_handle())
    return dx, dw, db


# ----------------------------------------------------------------------------------------------
# 1-channel map ops and losses
# ----------------------------------------------------------------------------------------------
def upsample_forward(x, scale: int):
    N, _, D, H, W = x.shape
    y = torch.empty((N, 1, D * scale, H * scale, W * scale), dtype=torch.float32, device=x.device)
    lib().call("pcrl_upsample_trilinear_fwd", x.contiguous(), y, N, D, H, W, scale, stream_handle())
    return y


def upsample_backward(dy, in_shape, scale: int):
    N, _, D, H, W = in_shape
    dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    lib().call("pcrl_upsample_trilinear_bwd", dy.contiguous(), dx, N, D, H, W, scale, stream_handle())
    return dx


def conv1x1_to1_forward(x, w, b, dtype):
    """OutputTransition: sigmoid(conv1x1x1(x)) -> (out [N,1,D,H,W] float32)."""
    L, s = lib(), stream_handle()
    N, D, H, W, C = dims(x)
    M = N * D * H * W
    K = w.shape[0]
    if K != 1:
        # n_class != 1 (constructor variant, models/pcrlv2_model_3d.py:78,109): one C -> 1 pass per class into a class-major buffer,
        # handed out as its [N, K, D, H, W] view
        wd, bd = w.detach(), b.detach()
        out = torch.empty((K, N, D, H, W), dtype=torch.float32, device=x.device)
        pre = _f32(M, x.device)
        for k in range(K):
            L.call("pcrl_conv3d_to1_fwd", x, wd[k], bd[k:k + 1], pre, None, None, 0, N, D, H, W, C, 1, dtype_code(dtype), s)
            L.call("pcrl_sigmoid_fwd", pre, out[k], M, s)
        return out.permute(1, 0, 2, 3, 4)
    pre = _f32(M, x.device)
    L.call("pcrl_conv3d_to1_fwd", x, w.detach(), b.detach(), pre, None, None, 0, N, D, H, W, C, 1, dtype_code(dtype), s)
    out = torch.empty((N, 1, D, H, W), dtype=torch.float32, device=x.device)
    L.call("pcrl_sigmoid_fwd", pre, out, M, s)
    return out


def conv1x1_to1_backward(x, out, dout, w, dtype, need_dx=True):
    L, s, dev = lib(), stream_handle(), x.device
    N, D, H, W, C = dims(x)
    M = N * D * H * W
    K = w.shape[0]
    if K != 1:
        outk, doutk = out.permute(1, 0, 2, 3, 4).contiguous(), dout.permute(1, 0, 2, 3, 4).contiguous()     # class-major, like the forward's buffer
        dw = torch.empty_like(w, dtype=torch.float32, memory_format=torch.contiguous_format)
        db = _f32(K, dev)
        dwk = dw.view(K, C)
        nb = L.call("pcrl_conv3d_to1_wgrad_ws_bytes", N, D, H, W, C, 1)
        dpre, dx, wd = _f32(M, dev), None, w.detach()
        for k in range(K):
            L.call("pcrl_sigmoid_bwd", doutk[k], outk[k], dpre, M, s)
            L.call("pcrl_conv3d_to1_wgrad", x, dpre, dwk[k], db[k:k + 1], workspace(nb, dev), nb, N, D, H, W, C, 1, dtype_code(dtype), s)
            if need_dx:
                nxt = torch.empty_like(x)
                L.call("pcrl_conv3d_to1_dgrad", dpre, wd[k], dx, nxt, N, D, H, W, C, 1, dtype_code(dtype), s)     # dx = sum over the classes
                dx = nxt
        return dx, dw, db
    dpre = _f32(M, dev)
    L.call("pcrl_sigmoid_bwd", dout.contiguous(), out, dpre, M, s)
    dw = torch.empty_like(w, dtype=torch.float32, memory_format=torch.contiguous_format)
    db = _f32(1, dev)
    nb = L.call("pcrl_conv3d_to1_wgrad_ws_bytes", N, D, H, W, C, 1)
    L.call("pcrl_conv3d_to1_wgrad", x, dpre, dw, db, workspace(nb, dev), nb, N, D, H, W, C, 1, dtype_code(dtype), s)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        L.call("pcrl_conv3d_to1_dgrad", dpre, w.detach(), None, dx, N, D, H, W, C, 1, dtype_code(dtype), s)
    return dx, dw, db


def concat_batch(tensors):
    """torch.cat(tensors, dim=0) of contiguous tensors of one shape / dtype / device as ONE launch (pcrl_concat: up to 8 pieces; the six local
    views of a step, train_3d.py:121) instead of one device copy per piece.  Falls back to torch.cat for what the kernel does not take."""
    t0 = tensors[0]
    ok = (1 <= len(tensors) <= 8 and t0.is_cuda and all(t.is_contiguous() and t.shape == t0.shape and t.dtype == t0.dtype and t.device == t0.device for t in tensors)
          and (t0.numel() * t0.element_size()) % 16 == 0 and all(t.data_ptr() % 16 == 0 for t in tensors))
    if not ok:
        return torch.cat(tensors, dim=0)
    import ctypes
    n = len(tensors)
    out = torch.empty((n * t0.shape[0],) + tuple(t0.shape[1:]), dtype=t0.dtype, device=t0.device)
    src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nb = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in tensors])
    lib().call("pcrl_concat", ctypes.addressof(src), ctypes.addressof(nb), n, out, stream_handle())
    return out


def mse_forward(p, gt):
    L = lib()
    n = p.numel()
    loss = _f32(1, p.device)
    nb = L.call("pcrl_reduce_ws_bytes", n)
    L.call("pcrl_mse_fwd", p, gt, loss, workspace(nb, p.device), nb, n, stream_handle())
    return loss.view(())


def mse_backward(p, gt, dloss):
    dp = torch.empty_like(p)
    lib().call("pcrl_mse_bwd", p, gt, dloss.contiguous().view(1), dp, p.numel(), stream_handle())
    return dp


def cosine_mean_forward(x, y, eps=1e-8):
    rows, C = x.shape
    out = _f32(1, x.device)
    saved = _f32(rows * 3, x.device)
    lib().call("pcrl_cosine_mean_fwd", x, y, out, saved, rows, C, eps, stream_handle())
    return out.view(()), saved


def cosine_mean_backward(x, y, saved, dout, eps=1e-8):
    rows, C = x.shape
    dx = torch.empty_like(x)
    lib().call("pcrl_cosine_mean_bwd", x, y, saved, dout.contiguous().view(1), dx, rows, C, eps, stream_handle())
    return dx


# ----------------------------------------------------------------------------------------------
# Optional extra (not in the reference): NT-Xent contrastive loss
# ----------------------------------------------------------------------------------------------
def ntxent_forward(z, tau, eps=1e-8):
    """z: float32 [2N, C] = [z1; z2].  -> (loss scalar tensor, saved workspace for the backward)."""
    R, C = z.shape
    L = lib()
    nb = L.call("pcrl_ntxent_ws_bytes", R, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=z.device)   # private: the backward reads what the forward left in it
    loss = _f32(1, z.device)
    L.call("pcrl_ntxent_fwd", z, loss, ws, nb, R, C, float(tau), float(eps), stream_handle())
    return loss.view(()), ws


def ntxent_backward(z, dloss, ws, tau, eps=1e-8):
    R, C = z.shape
    dz = torch.empty_like(z)
    lib().call("pcrl_ntxent_bwd", z, dloss.contiguous().view(1), dz, ws, ws.numel(), R, C, float(tau), float(eps), stream_handle())
    return dz


# ----------------------------------------------------------------------------------------------
# Optional extra (not in the reference): GroupNorm(G) + activation on NDHWC activations
# ----------------------------------------------------------------------------------------------
def gn_act_forward(y, gamma, beta, groups, act, dtype, eps=1e-5):
    """a = act(GroupNorm(groups)(y)); y, a: activations [N, C, D, H, W] in NDHWC storage.  -> (a, saved tuple for the backward)."""
    N, D, H, W, C = dims(y)
    a, saved = gn_forward_rows(y.permute(0, 2, 3, 4, 1).view(N, D * H * W, C), gamma, beta, groups, act, dtype, eps)
    return a.view(N, D, H, W, C).permute(0, 4, 1, 2, 3), saved


def gn_act_backward(da, saved, gamma, groups, act, dtype):
    """-> (dy, dgamma [C], dbeta [C])."""
    N, D, H, W, C = dims(da)
    dy, dg, db = gn_backward_rows(da.permute(0, 2, 3, 4, 1).view(N, D * H * W, C), saved, gamma, groups, act, dtype)
    return dy.view(N, D, H, W, C).permute(0, 4, 1, 2, 3), dg, db


def gn_forward_rows(y, gamma, beta, groups, act, dtype, eps=1e-5):
    """GroupNorm(groups) + activation on a contiguous [N, S, C] tensor (S rows of C channels per sample).  groups == C: InstanceNorm."""
    L, s, dev = lib(), stream_handle(), y.device
    N, S, C = y.shape
    tiles = L.call("pcrl_gn_stats_tiles", S)
    partial = _f32(N * tiles * C * 2, dev)
    L.call("pcrl_gn_stats", y, partial, N, S, C, dtype_code(dtype), s)
    mean_c, rstd_c, scale, shift = (_f32(N * C, dev) for _ in range(4))
    L.call("pcrl_gn_finalize", partial, tiles, N, S, C, groups, gamma, beta, float(eps), mean_c, rstd_c, scale, shift, s)
    a = torch.empty_like(y)
    for n in range(N):   # the streaming kernels take one coefficient row per launch
        L.call("pcrl_bn_act_apply", y[n], a[n], scale[n * C:], shift[n * C:], S, C, act, dtype_code(dtype), s)
    return a, (y, mean_c, rstd_c, scale, shift)


def gn_backward_rows(da, saved, gamma, groups, act, dtype):
    """-> (dy [N, S, C], dgamma [C], dbeta [C]) for gn_forward_rows."""
    y, mean_c, rstd_c, scale, shift = saved
    L, s, dev = lib(), stream_handle(), y.device
    N, S, C = y.shape
    rows_b = L.call("pcrl_bn_bwd_partial_rows", S)
    partial_b = _f32(N * rows_b * C * 2, dev)
    for n in range(N):
        L.call("pcrl_bn_act_bwd_reduce", da[n], y[n], scale[n * C:], shift[n * C:], mean_c[n * C:], rstd_c[n * C:],
               partial_b[n * rows_b * C * 2:], S, C, act, dtype_code(dtype), s)
    k1, kB, kA, dg_n, db_n = (_f32(N * C, dev) for _ in range(5))
    L.call("pcrl_gn_bwd_finalize", partial_b, rows_b, N, S, C, groups, gamma, mean_c, rstd_c, k1, kB, kA, dg_n, db_n, s)
    dy = torch.empty_like(y)
    for n in range(N):
        L.call("pcrl_bn_act_bwd_apply", da[n], y[n], dy[n], scale[n * C:], shift[n * C:], k1[n * C:], kB[n * C:], kA[n * C:], S, C, act,
               dtype_code(dtype), s)
    return dy, dg_n.view(N, C).sum(0), db_n.view(N, C).sum(0)

