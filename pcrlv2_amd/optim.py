"""Fused SGD over a flat parameter arena -- torch.optim.SGD as configured at train_3d.py:48-51
(momentum, weight decay on every parameter, dampening 0, no nesterov), one kernel launch per step.

`FusedSGD` subclasses torch.optim.SGD so `param_groups` (what utils.adjust_learning_rate edits) and
the checkpoint's `optimizer.state_dict()` layout ('momentum_buffer' per parameter) stay those of the
reference.  Parameters are re-homed into one contiguous float32 arena (values preserved; the
nn.Parameter objects stay the same), momentum buffers into a second arena, and gradients are
summed into a third (which is also the buffer the data-parallel all-reduce works on).
Parameters whose .grad is None are skipped, like torch.optim.SGD does after zero_grad(set_to_none=True).
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import lib, stream_handle


class FusedSGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, **kw):
        # the reference's argparse leaves --momentum / --weight_decay as strings when given on the CLI
        super().__init__(params, lr=float(lr), momentum=float(momentum), weight_decay=float(weight_decay), **kw)
        if len(self.param_groups) != 1:
            raise ValueError("FusedSGD supports a single param group (the reference uses one)")
        g = self.param_groups[0]
        if g["dampening"] != 0 or g["nesterov"] or g.get("maximize", False):
            raise ValueError("FusedSGD implements dampening=0, nesterov=False, maximize=False")
        self._plist = [p for p in g["params"]]
        dev = self._plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedSGD needs parameters on the GPU (no CPU fallback)")
        sizes = [p.numel() for p in self._plist]
        # every parameter starts on a 16-byte boundary of the arena (slots rounded up to 4 floats; the padding stays zero in all three
        # arenas): the kernels that read weights with 16-byte loads (the heads' Linear products) take them as they lie -- the model has
        # 1-element parameters (the deep-supervision heads' 1-channel convolutions and norms) in front of them
        self._slot_sizes = [(n + 3) // 4 * 4 for n in sizes]
        offs = [0]
        for n in self._slot_sizes:
            offs.append(offs[-1] + n)
        self._offsets_host = offs
        self._total = offs[-1]
        arenas = []
        for _ in range(3):         # zero-filled by the library's own entry point (a runtime memset): no ATen fill kernel even at set-up
            a = torch.empty(self._total, dtype=torch.float32, device=dev)
            lib().call("pcrl_zero", a, 4 * self._total, stream_handle())
            arenas.append(a)
        self.flat_p, self.flat_g, self.flat_buf = arenas
        with torch.no_grad():
            for p, o, n in zip(self._plist, offs, sizes):
                if p.dtype != torch.float32:
                    raise ValueError("FusedSGD: parameters must be float32")
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
        self._offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._gviews = [self.flat_g[o:o + n].view(p.shape) for p, o, n in zip(self._plist, offs, sizes)]
        self._numels = torch.tensor(sizes, dtype=torch.int64, device=dev)
        import ctypes
        self._offsets_c = (ctypes.c_int64 * len(offs))(*offs)       # pcrl_grad_sum reads the offsets on the host as well
        for i, (p, v) in enumerate(zip(self._plist, self._gviews)):
            p._pcrl_gview = v      # pcrlv2_amd.functions.flush_param_grads sums the step's gradients straight into the arena
            p._pcrl_gslot = (self, i)
        self._initialised = [False] * len(self._plist)
        self._flag_cache = {}
        self._flag_ring = None
        self.grad_scale = 1.0          # set to 1/world_size by the data-parallel wrapper
        self.pre_step = None           # optional callable(self, None) -> has-grad list: replaces the gradient gather (ddp.DataParallel)
        self.skip_flag = None          # float32[1] on the device, consumed by the NEXT step(): non-zero = leave parameters and momentum untouched
        ops.bump_weights_epoch()

    # ------------------------------------------------------------------
    def gather_grads(self):
        """Gradients that do not already live in the flat arena (set by hand, or produced by autograd's own accumulation)
        are copied into it; returns the has-grad list."""
        has = [p.grad is not None for p in self._plist]
        todo = [(v, p.grad) for v, p, h in zip(self._gviews, self._plist, has) if h and p.grad.data_ptr() != v.data_ptr()]
        if todo:
            torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])
        return has

    def _flags(self, has):
        """int32 per parameter (bit 0: has a gradient, bit 1: momentum buffer initialised) on the device.  Patterns are cached; a NEW pattern
        (the 2D step has 160 of them: which scales the 13 draws picked) is staged in a pinned host slot and copied asynchronously on the
        current stream -- `torch.tensor(vals, device=...)` is a pageable-memory copy, which the runtime completes only after everything
        queued before it: a host stall of the whole backward (14 ms per 2D step, tools/host_probe_2d.py).  A slot is reused after
        `len(ring)` further misses; the host never runs more than config.MAX_STEPS_AHEAD steps ahead of the device, and the ring is deeper."""
        key = (tuple(has), tuple(self._initialised))
        t = self._flag_cache.get(key)
        if t is None:
            vals = [(1 if h else 0) | (2 if i else 0) for h, i in zip(has, self._initialised)]
            if self._flag_ring is None:
                self._flag_ring = [torch.empty(len(vals), dtype=torch.int32).pin_memory() for _ in range(8)]
                self._flag_ring_ev = [None] * 8
                self._flag_ring_pos = 0
            k = self._flag_ring_pos = (self._flag_ring_pos + 1) % len(self._flag_ring)
            if self._flag_ring_ev[k] is not None:
                self._flag_ring_ev[k].synchronize()       # the copy that last read this slot (long done: 8 misses ago)
            self._flag_ring[k].copy_(torch.tensor(vals, dtype=torch.int32))
            t = self._flag_ring[k].to(self.flat_p.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._flag_ring_ev[k] = ev
            if len(self._flag_cache) > 256:
                self._flag_cache.clear()
            self._flag_cache[key] = t
        return t

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """torch.optim.SGD layout in ('momentum_buffer' per parameter, e.g. from a checkpoint written by the reference or by
        this class, train_3d.py:71-82); the buffers are copied into the flat arena the kernel updates."""
        super().load_state_dict(state_dict)
        if len(self.param_groups) != 1:
            raise ValueError("FusedSGD supports a single param group")
        for i, p in enumerate(self._plist):
            buf = self.state.get(p, {}).get("momentum_buffer")
            o, n = self._offsets_host[i], p.numel()
            view = self.flat_buf[o:o + n].view(p.shape)
            if buf is None:
                view.zero_()
                self._initialised[i] = False
                self.state.pop(p, None)
            else:
                view.copy_(buf.to(device=view.device, dtype=torch.float32))
                self.state[p]["momentum_buffer"] = view
                self._initialised[i] = True
        self._flag_cache.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        ops.join_side_stream()
        if self.pre_step is not None:
            has = self.pre_step(self, None)       # the data-parallel wrapper gathers (bucket by bucket) and all-reduces
        else:
            has = self.gather_grads()
        flags = self._flags(has)
        skip, self.skip_flag = self.skip_flag, None
        if skip is None:
            lib().call("pcrl_sgd_step", self.flat_p, self.flat_g, self.flat_buf, self._offsets, flags, len(self._plist),
                       self._total, float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), float(self.grad_scale),
                       stream_handle())
        else:
            # the divergence guard decided on the device (train_3d.train_step): a skipped update leaves the arenas bit-unchanged.  The host-side
            # "momentum initialised" marks below are set either way; the buffers start as zeros, so momentum * 0 + g == g and a first update
            # that happens one step later is the same arithmetic.
            lib().call("pcrl_sgd_step_guarded", self.flat_p, self.flat_g, self.flat_buf, self._offsets, flags, len(self._plist),
                       self._total, float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), float(self.grad_scale), skip,
                       stream_handle())
        for i, (p, h) in enumerate(zip(self._plist, has)):
            if h and not self._initialised[i]:
                self._initialised[i] = True
                o, n = self._offsets_host[i], p.numel()
                self.state[p]["momentum_buffer"] = self.flat_buf[o:o + n].view(p.shape)
        ops.bump_weights_epoch()
        ops.begin_step()
        return loss
