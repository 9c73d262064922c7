"""Data parallelism the MI355X way: one process per GPU, gradient all-reduce over RCCL/xGMI.

Replaces the reference's single-process `nn.DataParallel` (train_3d.py:54): no per-forward
parameter broadcast, no scatter/gather of activations -- each rank runs the whole step on its own
b crops and only the 68 MB of gradients cross xGMI once per step, as a few large buckets
(ring all-reduce is per-link bound on xGMI, so few large messages beat many small ones).
BatchNorm statistics stay per rank, like the reference's per-replica statistics.

`BucketedAllReduce` is device-agnostic (tested with gloo on CPU, world_size 2); on the GPU the
collectives run on a side stream so that they overlap whatever the main stream still has queued
(the tail of backward: kernels are launched asynchronously, the host gets here early).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist



def init_process_group_from_env(backend: str | None = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", 0))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:     # PCRL_DIST_BACKEND=gloo: several ranks on ONE GPU (tests of the N-rank entry points on a one-GPU box)
        backend = os.environ.get("PCRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" IS RCCL on ROCm
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1:
        # The (host, NUMA node, rank) exchange is a COLLECTIVE: every rank reaches it, whatever its own PCRL_BIND_CPUS says and whether or not it
        # can read its node (a rank that opted out, or raised before the gather, left its peers blocked inside an optional optimisation).
        table = None
        try:
            import socket
            node = _gpu_numa_node(torch.cuda.current_device()) if torch.cuda.is_available() else -1
        except Exception:
            node = -1
        try:
            table = [None] * world
            dist.all_gather_object(table, (socket.gethostname(), node, rank))
        except Exception as e:      # a failing collective is a broken group, not a binding problem: say so and carry on unbound
            print(f"[pcrlv2_amd.ddp] rank table exchange failed ({e}); CPU binding skipped", flush=True)
            table = None
        if table is not None:
            bind_rank_to_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), verbose=os.environ.get("PCRL_BIND_VERBOSE", "0") == "1", table=table)
    return rank, world, local


def shutdown(ok: bool = True):
    """Leave the process group the way every N-rank entry point must (bench.py, main.py -> train_3d / train_2d): a barrier so that no rank tears
    its transport down under a peer's last collective, then `destroy_process_group()`.  The reference's nn.DataParallel (train_3d.py:54) lives
    in one process and needs none of this; one process per GPU does -- a rank that returns from main() with the group alive leaves gloo's /
    RCCL's worker threads running into interpreter teardown, which now and then ends in `terminate called without an active exception`
    (SIGABRT, torchrun reports rc = 1 although every result was already printed).  `ok=False` (an exception is propagating): no barrier --
    the peers may never reach it -- only the teardown.  A no-op without a group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if ok:
        try:
            if torch.cuda.is_available() and dist.get_backend() == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
        except Exception as e:      # a broken group must not turn a finished run into a failed one
            print(f"[pcrlv2_amd.ddp] barrier before shutdown failed ({e})", flush=True)
    try:
        dist.destroy_process_group()
    except Exception as e:
        print(f"[pcrlv2_amd.ddp] destroy_process_group failed ({e})", flush=True)


def parse_cpulist(text: str):
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def cpu_share(node_cpus, allowed, peers_on_node: int, my_slot: int):
    """The CPUs of one rank: the GPU's NUMA node's CPUs that this process may use, split evenly among the `peers_on_node` ranks whose GPUs
    hang off the same node (slot = this rank's position among them).  Falls back to an even split of everything allowed."""
    pool = [c for c in node_cpus if c in allowed] or sorted(allowed)
    n = max(1, len(pool) // max(peers_on_node, 1))
    mine = pool[my_slot * n:(my_slot + 1) * n]
    return mine or pool


def _gpu_numa_node(index: int):
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read().strip())
    except Exception:
        return -1


WORKER_CPUS = None      # set by bind_rank_to_numa: the CPUs this rank's DataLoader workers should run on (data.worker_affinity_init applies it)


def split_share(mine, workers: int):
    """A rank's CPU share -> (launcher CPUs, loader-worker CPUs).  The launcher thread (it enqueues ~1 000 kernel launches per step: SURVEY 8e, the
    >= 6x target is bounded by its jitter) keeps CPUs of its own; the `workers` DataLoader processes get the rest, one CPU each where the share
    allows.  A share too small for that (fewer than workers + 2 CPUs) is not split: everything runs on all of it (and the log says so)."""
    mine = list(mine)
    if workers <= 0 or len(mine) < workers + 2:
        return mine, mine
    k = max(2, len(mine) - workers)
    return mine[:k], mine[k:]


def bind_rank_to_numa(local_rank: int, local_world: int, verbose: bool = False, exchange: bool = False, table=None, workers=None):
    """Pin this rank's host threads to CPUs of its GPU's NUMA node (at < 1 ms of communication per ~33 ms step the >= 6x target
    at 8 GPUs is bounded by host-side launch jitter, not by xGMI: eight launcher threads migrating across two sockets is that jitter).
    The node is that of the device this process has made CURRENT (`torch.cuda.current_device()` after `set_device`: correct under
    HIP_VISIBLE_DEVICES remapping, where local device index != local rank); `table` = the [(host, node, rank)] list every rank contributed to
    (init_process_group_from_env gathers it unconditionally) lets ranks that share a node split its CPUs -- without it the split assumes device i
    belongs to local rank i.  (`exchange=True` without a table gathers it here: kept for callers that build their own group.)
    The rank's share is split again (split_share): the LAUNCHER thread is pinned to the first part, the DataLoader workers (`workers`, default
    $PCRL_LOADER_WORKERS or main.py's --workers) are told to run on the rest through ddp.WORKER_CPUS / $PCRL_WORKER_CPUS, which
    data.worker_affinity_init applies in each worker (a forked worker inherits the launcher's narrow mask; it widens itself) -- round 4 left the
    workers inside the launcher's mask, time-slicing with it.  Logged ONCE on rank 0 (stderr), with the split.  Also `torch.set_num_threads(<= 8)`.
    PCRL_BIND_CPUS=0 turns the binding off (after the exchange).  -> the launcher's CPU list, or None when nothing was done."""
    global WORKER_CPUS
    if os.environ.get("PCRL_BIND_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import socket
        allowed = os.sched_getaffinity(0)
        cuda = torch.cuda.is_available()
        node = _gpu_numa_node(torch.cuda.current_device()) if cuda else -1
        how = "current device"
        if table is None and exchange and dist.is_available() and dist.is_initialized():
            table = [None] * dist.get_world_size()
            dist.all_gather_object(table, (socket.gethostname(), node, dist.get_rank()))
        if table is not None:
            me = dist.get_rank() if dist.is_available() and dist.is_initialized() else local_rank
            mates = sorted(r for (h, n, r) in table if h == socket.gethostname() and n == node)
            peers, slot = len(mates), mates.index(me)
        else:
            nodes = [_gpu_numa_node(i) for i in range(local_world)] if cuda and torch.cuda.device_count() >= local_world else [-1] * local_world
            if not cuda:
                node = nodes[local_rank]
            mates = [r for r in range(local_world) if nodes[r] == node] or [local_rank]
            peers, slot = len(mates), (mates.index(local_rank) if local_rank in mates else 0)
            how = "device index = local rank assumed for the peers"
        node_cpus = sorted(allowed)
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                node_cpus = parse_cpulist(f.read())
        mine = cpu_share(node_cpus, allowed, peers, slot)
        if workers is None:
            workers = int(os.environ.get("PCRL_LOADER_WORKERS", "0") or 0)
        launcher, wcpus = split_share(mine, workers)
        os.sched_setaffinity(0, launcher)
        WORKER_CPUS = list(wcpus)
        os.environ["PCRL_WORKER_CPUS"] = ",".join(str(c) for c in wcpus)
        os.environ["PCRL_WORKER_CPUS_SPLIT"] = "1" if list(wcpus) != list(launcher) else "0"   # unsplit share: the workers share it, unpinned
        torch.set_num_threads(max(1, min(8, len(launcher))))
        if verbose or (dist.is_available() and dist.is_initialized() and dist.get_rank() == 0):
            import sys
            split = (f"launcher thread on {len(launcher)} ({launcher[0]}..{launcher[-1]}), {workers} loader workers on {len(wcpus)} ({wcpus[0]}..{wcpus[-1]})"
                     if wcpus is not launcher and wcpus != launcher else
                     f"NOT split: {workers} loader workers share the launcher's {len(mine)} CPUs ({mine[0]}..{mine[-1]})" if workers > 0 else
                     f"launcher on all {len(mine)} ({mine[0]}..{mine[-1]}); no loader workers announced (PCRL_LOADER_WORKERS)")
            print(f"[pcrlv2_amd.ddp] rank binding on ({how}): local rank {local_rank} -> GPU NUMA node {node}, share {len(mine)} CPUs: {split}; "
                  f"{torch.get_num_threads()} torch threads; PCRL_BIND_CPUS=0 disables", file=sys.stderr, flush=True)
        return launcher
    except Exception as e:      # binding is an optimisation, never a reason to fail a run
        if verbose:
            print(f"[pcrlv2_amd.ddp] CPU binding skipped: {e}", flush=True)
        return None


# ---- opt-in: nn.DataParallel's LITERAL partition of the local views (PCRL_DP_LOCAL_PARTITION=chunk) -------------------------------------------
# The reference concatenates the six local views of the GLOBAL batch view-major into one [6B] tensor and lets DataParallel scatter THAT along dim 0
# (train_3d.py:121-123): replica r runs rows [r * 6B / W, (r + 1) * 6B / W) -- with two replicas, replica 0 sees local views 0-2 of ALL samples -- so the
# BatchNorm statistics of the local pass group rows differently than when every rank keeps the six views of its own crops (this engine's default and
# one stated deviation).  This mode reproduces the reference's grouping at the price of two small exchanges on the data path per step: the local
# views' inputs (6 b x 16 KiB per rank) before the local forward and their pooled features ([6 b, 896] floats) after it, plus the features' gradient
# in backward -- all-gathers / one all-reduce (gloo has no all-to-all), nothing a pre-training run would want by choice.  Pinned by
# tests/golden/dp2chunk_b4x2_32x32x16.npz (the REAL model under the literal scatter; oracle/make_golden.py --data-parallel).
def chunk_partition_on() -> bool:
    return os.environ.get("PCRL_DP_LOCAL_PARTITION", "sample") == "chunk" and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def chunk_local_inputs(loc: torch.Tensor, b: int, nl: int) -> torch.Tensor:
    """loc = this rank's local views, view-major [nl * b, ...] -> this rank's chunk of the GLOBAL view-major tensor (same number of rows)."""
    W, r = dist.get_world_size(), dist.get_rank()
    loc = loc.contiguous()
    parts = [torch.empty_like(loc) for _ in range(W)]
    dist.all_gather(parts, loc)
    G = torch.stack(parts).view(W, nl, b, *loc.shape[1:]).transpose(0, 1).reshape(nl * W * b, *loc.shape[1:])     # row v * B + rank * b + i
    rows = nl * b
    return G[r * rows:(r + 1) * rows].contiguous()


class _OwnRowsOfGathered(torch.autograd.Function):
    """x = the features this rank computed for ITS chunk of the global local-view tensor ([nl * b, C]) -> the features of this rank's OWN samples,
    view-major ([nl * b, C]: DataParallel's gather followed by `t[:, bsz * i: bsz * (i + 1)]` restricted to the rank's samples).  Backward: the
    gradient rows go back to the ranks that computed them (all-reduce of the scattered gradient, then this rank's chunk)."""

    @staticmethod
    def forward(ctx, x, b, nl):
        W, r = dist.get_world_size(), dist.get_rank()
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(W)]
        dist.all_gather(parts, x)
        G = torch.cat(parts, dim=0).view(nl, W, b, x.shape[1])          # the chunks in rank order ARE the global row order
        ctx.geom = (W, r, b, nl, x.shape[1])
        return G[:, r].reshape(nl * b, x.shape[1]).contiguous()

    @staticmethod
    def backward(ctx, g):
        W, r, b, nl, C = ctx.geom
        full = torch.zeros(nl, W, b, C, dtype=g.dtype, device=g.device)
        full[:, r] = g.reshape(nl, b, C)
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
        rows = nl * b
        return full.view(nl * W * b, C)[r * rows:(r + 1) * rows].contiguous(), None, None


def chunk_local_features(feats_loc, b: int, nl: int):
    """feats_loc = [[projection, prediction] per scale] of the rank's CHUNK -> the same structure for the rank's OWN samples (one exchange for all six)."""
    flat = [t for pair in feats_loc for t in pair]
    widths = [t.shape[1] for t in flat]
    own = _OwnRowsOfGathered.apply(torch.cat([t.float() for t in flat], dim=1), b, nl)
    out, o = [], 0
    for k in range(len(feats_loc)):
        pair = []
        for _ in range(2):
            w = widths[len(out) * 2 + len(pair)]
            pair.append(own[:, o:o + w].contiguous())
            o += w
        out.append(pair)
    return out


def plan_buckets(sizes, bucket_elems: int):
    """Contiguous element ranges [(begin, end)] over the flat arena, built from the LAST parameter backwards
    (gradients become final in reverse execution order), each at most `bucket_elems` unless one tensor is larger."""
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    buckets, end, i = [], offs[-1], len(sizes)
    while i > 0:
        j = i
        while j > 0 and (offs[i] - offs[j - 1] <= bucket_elems or j == i):
            j -= 1
        buckets.append((offs[j], offs[i]))
        i = j
    assert buckets[0][1] == end and buckets[-1][0] == 0
    return buckets


class BucketedAllReduce:
    def __init__(self, flat_g: torch.Tensor, sizes, group=None, bucket_mb: float = 24.0):
        self.flat_g = flat_g
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = plan_buckets(list(sizes), int(bucket_mb * (1 << 20) / 4))
        self.comm_stream = torch.cuda.Stream(device=flat_g.device) if flat_g.is_cuda else None

    def reduce(self):
        """SUM all-reduce of the flat gradient arena, bucket by bucket (last parameters first)."""
        if self.world == 1 and not dist.is_initialized():
            return
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                works = [dist.all_reduce(self.flat_g[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                         for b, e in self.buckets]
                for w in works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            works = [dist.all_reduce(self.flat_g[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                     for b, e in self.buckets]
            for w in works:
                w.wait()


class DataParallel:
    """Ties a FusedSGD-style optimizer to the process group: parameter/buffer broadcast at start; per step a SUM
    all-reduce of the flat gradient arena with 1/world folded into the SGD kernel.

    Overlap with backward.  A step runs three forwards (view 1, view 2, local views) that share all parameters, and
    autograd replays them in reverse creation order: local, view 2, view 1.  A parameter's gradient is therefore FINAL
    once the stage of the FIRST forward of the step (pass 0) has run its backward.  The stage Functions park their parameter
    gradients (pcrlv2_amd.functions._park) and report final parameters through `functions.mark_final`; as soon as every
    parameter of a bucket is final the bucket's parked gradients are summed into the flat arena and its all-reduce is launched
    on the side stream -- while the rest of backward (earlier layers) is still running.  Whatever is not final by then
    (parameters without a pass-0 gradient, e.g. the unused deep-supervision heads) is swept up in optimizer.step().
    A gradient that arrives for a bucket already sent ("late"; cannot happen with the reference's step) is all-reduced on its
    own and added.

    Default: `overlap=False` (PCRL_DDP_OVERLAP=1 turns it on) -- a MEASURED choice (tools/ddp_overlap_probe.sh: the wrapper on a one-rank
    RCCL group on one MI355X, where the collectives themselves cost nothing): launching buckets from inside backward costs 1.2-1.7 ms
    per step (34.2-34.6 vs 32.4-32.7 ms), launching them all from optimizer.step() costs nothing measurable (33.1-33.3 vs 33.3 without a
    wrapper).  The reason is structural: a bucket is made of WEIGHT gradients, those are produced on the side stream, and the side stream is
    the one that finishes last (it carries the ~10 ms backlog of weight-gradient kernels that run next to the data-gradient chain) -- in the
    rocprofv3 trace the first bucket's sums execute at 34.2 ms of a 38 ms step however early they were queued.  There is nothing to overlap
    the 68 MB all-reduce with except that backlog, and all-reducing at the end costs its own time only (~0.7 ms on 8 GPUs at RCCL's
    large-message rate).  Either way the collectives and the bucket sums run on the side stream, behind the weight gradients they
    consume, so they start the moment those are done.
    """

    def __init__(self, model: torch.nn.Module, optimizer, group=None, bucket_mb: float = 24.0, strict_flags: bool = False,
                 overlap=None, force_collectives: bool = False):
        from . import functions as Fn
        self._fn = Fn
        self.model, self.opt, self.group = model, optimizer, group
        self.strict_flags = strict_flags
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._active = self.world > 1 or (force_collectives and dist.is_initialized())   # 1-rank groups: test hook only
        self._sizes = list(getattr(optimizer, "_slot_sizes", None) or [p.numel() for p in optimizer._plist])   # arena slots (FusedSGD pads to 16 bytes)
        optimizer.grad_scale = 1.0 / self.world
        optimizer.pre_step = self._pre_step
        optimizer.data_parallel = self          # train_3d.train_step reduces the divergence flag on this wrapper's group
        self._index = {id(p): i for i, p in enumerate(optimizer._plist)}
        for p, v in zip(optimizer._plist, optimizer._gviews):
            p._pcrl_gview = v
        self.configure((os.environ.get("PCRL_DDP_OVERLAP", "0") == "1") if overlap is None else overlap, bucket_mb)
        self.broadcast_state()

    def configure(self, overlap: bool, bucket_mb: float = 24.0):
        """(Re)plan the buckets and choose where they are launched from: inside backward as they become final (`overlap`) or all from
        optimizer.step().  Callable between steps (bench.py's A/B of the four settings); the result of a step does not depend on it."""
        optimizer = self.opt
        self.bucket_mb = float(bucket_mb)
        self.reducer = BucketedAllReduce(optimizer.flat_g, self._sizes, self.group, bucket_mb)
        if optimizer.flat_g.is_cuda and os.environ.get("PCRL_DDP_COMM_STREAM", "side") == "side":
            # bucket sums and collectives on the engine's side stream rather than on a stream of their own: main, view, side and RCCL's
            # internal stream are then the four streams of a rank -- as many as ROCm's default hardware queues (measured on one GPU with a
            # one-rank RCCL group: a fifth stream costs 1.7 ms per step; GPU_MAX_HW_QUEUES=8 costs 7.5 ms once RCCL is initialised)
            from . import ops
            self.reducer.comm_stream = ops.side_stream(optimizer.flat_g.device)
        self.overlap = bool(overlap)
        sizes = self._sizes
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        self._bucket_params = []
        for (b, e) in self.reducer.buckets:
            self._bucket_params.append([i for i in range(len(sizes)) if offs[i] >= b and offs[i + 1] <= e])
        self._param_bucket = {}
        for bi, idxs in enumerate(self._bucket_params):
            for i in idxs:
                self._param_bucket[i] = bi
        self._reset_step()
        if self._active and self.overlap:
            self._fn.set_ddp_callbacks(self._on_final, self._on_backward_end)
        else:
            self._fn.set_ddp_callbacks(None, None)     # a wrapper built earlier in this process must not keep receiving callbacks
        return self

    # ------------------------------------------------------------------
    def _reset_step(self):
        self._ready = [0] * len(self.reducer.buckets)
        self._launched = [False] * len(self.reducer.buckets)
        self._gathered = [False] * len(self.opt._plist)
        self._final = [False] * len(self.opt._plist)
        self._works = []
        self._late = []            # (parameter index, summed late gradient)

    def _on_final(self, p):
        """functions.mark_final: nothing will add to this parameter's gradient any more in this step."""
        i = self._index.get(id(p))
        if i is None or self._final[i] or self._gathered[i]:
            return
        self._final[i] = True
        bi = self._param_bucket[i]
        self._ready[bi] += 1
        if self._ready[bi] == len(self._bucket_params[bi]) and not self._launched[bi]:
            self._launch_bucket(bi)

    @torch.no_grad()
    def _on_backward_end(self):
        """End of a backward pass: gradients still parked go to `.grad`; those of buckets already sent are kept aside."""
        Fn = self._fn
        from . import ops
        ops.join_side_stream()
        late = []
        if any(self._launched):
            for p in Fn.parked_params():
                i = self._index.get(id(p))
                if i is not None and self._gathered[i]:
                    late.append(p)
        for p in late:
            gs = Fn.take_parked(p)
            acc = gs[0].clone()
            for g in gs[1:]:
                acc += g
            self._late.append((self._index[id(p)], acc))
        Fn.flush_param_grads()

    @torch.no_grad()
    def _launch_bucket(self, bi):
        """Sum the bucket's gradients into the flat arena (zeros for parameters without one) and start its all-reduce."""
        opt = self.opt
        idxs = self._bucket_params[bi]
        cs = self.reducer.comm_stream
        b, e = self.reducer.buckets[bi]
        seg = opt.flat_g[b:e]
        from . import ops
        import contextlib
        with ops.trace_range("all_reduce bucket %d (%.1f MB)" % (bi, (e - b) * 4 / 2**20)):
            # On the GPU everything a bucket needs -- summing its parked gradients into the arena, zero-filling parameters without a gradient,
            # the collective -- runs on the COMMUNICATION stream, which waits for the producers; the main stream is never joined here, so the
            # weight gradients queued on the side stream keep overlapping the data-gradient chain while buckets go out.
            self._fn.flush_param_grads([opt._plist[i] for i in idxs], on_stream=cs)
            if cs is not None:
                cs.wait_stream(torch.cuda.current_stream())
            with (torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext()):
                miss = [i for i in idxs if opt._plist[i].grad is None]
                copy = [i for i in idxs if opt._plist[i].grad is not None and opt._plist[i].grad.data_ptr() != opt._gviews[i].data_ptr()]
                if seg.is_cuda:
                    # no ATen launch on the step path: grad-less parameters are zeroed by pcrl_zero (a runtime memset) --
                    # consecutive ones (the unused deep-supervision heads sit next to each other in the arena) as ONE range
                    L, sh = ops.lib(), ops.stream_handle()
                    for i in copy:      # never on the training step's path (its gradients are parked and summed straight into the arena by flush_param_grads)
                        opt._gviews[i].copy_(opt._plist[i].grad)
                    offs = opt._offsets_host
                    k = 0
                    while k < len(miss):
                        j = k
                        while j + 1 < len(miss) and miss[j + 1] == miss[j] + 1:
                            j += 1
                        lo, hi = offs[miss[k]], offs[miss[j] + 1]
                        L.call("pcrl_zero", opt.flat_g[lo:hi], 4 * (hi - lo), sh)
                        k = j + 1
                else:       # CPU tensors (the gloo tests of this logic)
                    if copy:
                        torch._foreach_copy_([opt._gviews[i] for i in copy], [opt._plist[i].grad for i in copy])
                    if miss:
                        torch._foreach_zero_([opt._gviews[i] for i in miss])
                self._works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for i in idxs:
            self._gathered[i] = True
        self._launched[bi] = True

    def broadcast_state(self):
        if not self._active:
            return
        dist.broadcast(self.opt.flat_p, src=0, group=self.group)
        if getattr(self.opt, "flat_buf", None) is not None:        # momentum buffers of a resumed run
            dist.broadcast(self.opt.flat_buf, src=0, group=self.group)
        for b in self.model.buffers():
            dist.broadcast(b, src=0, group=self.group)

    def _pre_step(self, opt, has):
        """Runs inside optimizer.step() INSTEAD of the optimizer's own gradient gather (returns the has-grad list)."""
        has = [p.grad is not None for p in opt._plist]
        if not self._active:
            opt.gather_grads()
            return has
        if self.strict_flags:
            # A parameter is updated if ANY rank produced a gradient for it.  All ranks draw the loss scales from
            # identically seeded `random` streams (train_3d.seed_everything), so the pattern is the same everywhere and
            # this exchange (which costs a device sync) is a debug check, off by default.
            flags = torch.tensor([1 if h else 0 for h in has], dtype=torch.int32, device=opt.flat_g.device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            has = [bool(v) for v in flags.tolist()]
        for bi in range(len(self.reducer.buckets)):      # buckets are ordered last-parameters-first
            if not self._launched[bi]:
                self._launch_bucket(bi)
        cs = self.reducer.comm_stream
        if cs is not None:
            with torch.cuda.stream(cs):
                for w in self._works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(cs)
        else:
            for w in self._works:
                w.wait()
        if self._late:
            # a gradient arrived after its bucket was sent: by linearity, all-reduce the late part on its own and add it
            if not getattr(self, "_warned_late", False):
                print("[pcrlv2_amd.ddp] warning: gradient accumulation after a bucket was sent; reducing the late part separately")
                self._warned_late = True
            for i, acc in self._late:
                dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
                opt._gviews[i].add_(acc.view_as(opt._gviews[i]))
        self._reset_step()
        return has
