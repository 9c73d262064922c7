"""ctypes binding of libpcrl_hip.so.

Signatures are parsed from include/pcrl_hip.h (the single source of truth for the C ABI), so every
declared entry point is bound with exact argument types, and a missing symbol is an import-time
error.  There is NO fallback: if the library is absent or a call fails, a RuntimeError is raised
(the product path never routes through PyTorch eager ops or the CPU oracle).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import re
import threading

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), "include", "pcrl_hip.h")
# PCRL_LIB: another build of the same ABI (tools/ A/B probes only: a previous revision's library next to the current one)
LIBPATH = os.environ.get("PCRL_LIB") or os.path.join(_PKG, "lib", "libpcrl_hip.so")

PCRL_F32, PCRL_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU = 0, 1, 2, 3   # ACT_SILU: optional extra (GroupNorm+SiLU), not on the reference path
ACT_ELU = 4       # LUConv(act='elu'): a constructor variant of the reference (models/pcrlv2_model_3d.py:24-25)
CONV_BM = 128

_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
    "double": ctypes.c_double, "pcrl_stream_t": ctypes.c_void_p,
}
_RET = {"void": None, "int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}


def parse_header(path: str = HEADER):
    """-> {name: (return type string, [(arg type string, arg name), ...])} for every prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int64_t|size_t|int|void)\s+(pcrl_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        lst = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                lst.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, lst)
    return protos


def _ctype_of(t: str):
    if "*" in t:
        return ctypes.c_void_p
    return _CTYPES[t]


class PcrlError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIBPATH):
            raise PcrlError(
                f"{LIBPATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m pcrlv2_amd.build`).  pcrlv2_amd has no CPU / eager fallback.")
        self.cdll = ctypes.CDLL(LIBPATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (ret, args) in self.protos.items():
            try:
                f = getattr(self.cdll, name)
            except AttributeError as e:
                raise PcrlError(f"libpcrl_hip.so does not export {name} declared in pcrl_hip.h") from e
            f.restype = _RET[ret]
            f.argtypes = [_ctype_of(t) for t, _ in args]
            self.fn[name] = (f, ret, args)
        self.profiler = None
        self.counter = None
        self._kinds = {}
        self.debug_set_wgrad_tr = self.cdll.pcrl_debug_set_wgrad_tr
        self.debug_set_wgrad_tr.argtypes = [ctypes.c_int]
        self.debug_set_wgrad_tr.restype = None
        self.debug_set_wgrad_impl = self.cdll.pcrl_debug_set_wgrad_impl
        self.debug_set_wgrad_impl.argtypes = [ctypes.c_int]
        self.debug_set_wgrad_impl.restype = None
        self.debug_set_conv_impl = self.cdll.pcrl_debug_set_conv_impl
        self.debug_set_conv_impl.argtypes = [ctypes.c_int]
        self.debug_set_conv_impl.restype = None
        self.debug_set_conv2d_impl = self.cdll.pcrl_debug_set_conv2d_impl
        self.debug_set_conv2d_impl.argtypes = [ctypes.c_int]
        self.debug_set_conv2d_impl.restype = None

    def version(self) -> str:
        return self.fn["pcrl_version"][0]().decode()

    def last_error(self) -> str:
        return self.fn["pcrl_last_error"][0]().decode()

    @contextlib.contextmanager
    def count_calls(self, *names):
        """Counts the calls of the named entry points inside the block (tests: which kernels a step really used) -> {name: calls}."""
        class _Counter:
            def __init__(self, watch):
                self.watch, self.n = set(watch), {}

            def add(self, name, args):
                self.n[name] = self.n.get(name, 0) + 1

        prev, self.counter = self.counter, _Counter(names)
        try:
            yield self.counter.n
        finally:
            self.counter = prev

    def call(self, name: str, *args):
        """Call an `int pcrl_*` entry point; tensors -> device pointers, None -> NULL; raises on error."""
        f, ret, protos = self.fn[name]
        kinds = self._kinds.get(name)
        if kinds is None:       # 0: pointer, 1: stream handle, 2: float, 3: integer -- classified once per entry point
            kinds = self._kinds[name] = tuple(0 if "*" in t else 1 if t == "pcrl_stream_t" else 2 if t in ("float", "double") else 3 for t, _ in protos)
        if len(args) != len(kinds):
            raise TypeError(f"{name}: expected {len(protos)} arguments, got {len(args)}")
        conv = []
        add = conv.append
        for a, k in zip(args, kinds):
            if k == 0:
                if a is None:
                    add(None)
                elif isinstance(a, torch.Tensor):
                    if not a.is_cuda:
                        an = protos[len(conv)][1]
                        raise PcrlError(f"{name}: argument `{an}` is a {a.device} tensor; libpcrl_hip needs device memory (no CPU fallback)")
                    add(a.data_ptr())
                else:
                    add(int(a))
            elif k == 3:
                add(int(a))
            elif k == 2:
                add(float(a))
            else:
                add(a)
        if self.counter is not None and name in self.counter.watch:
            self.counter.add(name, args)       # bench.py: algorithmic bytes / flops of the launches of a region (no events, no timing)
        if self.profiler is not None and name in self.profiler.watch:
            r = self.profiler.timed(name, args, f, conv)
        else:
            r = f(*conv)
        if ret == "int" and r != 0:
            raise PcrlError(f"{name} failed ({r}): {self.last_error()}")
        return r


class EventProfiler:
    """Times selected entry points with HIP events recorded on the launch stream (torch's current stream, which is
    the stream every pcrl_* call is given).  `keyfn(name, args) -> (key, work)` classifies a launch and returns its
    algorithmic work (flops or bytes); results: {key: [launches, total_ms, total_work]}."""

    def __init__(self, watch, keyfn):
        self.watch, self.keyfn = set(watch), keyfn
        self.pending = []

    def timed(self, name, args, f, conv):
        key, work = self.keyfn(name, args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*conv)
        e1.record()
        self.pending.append((key, work, e0, e1))
        return r

    def results(self):
        torch.cuda.synchronize()
        out = {}
        for key, work, e0, e1 in self.pending:
            rec = out.setdefault(key, [0, 0.0, 0.0])
            rec[0] += 1
            rec[1] += e0.elapsed_time(e1)
            rec[2] += work
        return out


_lib = None
_lock = threading.Lock()


def lib() -> _Lib:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = _Lib()
    return _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def stream_handle() -> int:
    """hipStream_t of torch's current stream on the current device.  A training step asks ~1000 times: the raw accessor (what torch's own
    compiled graphs use) costs 0.3 us where `torch.cuda.current_stream().cuda_stream` builds a Stream object each time (11 us: 11 ms of host
    time per 2D step, tools/host_probe_2d.py)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return PCRL_F32
    if dt == torch.bfloat16:
        return PCRL_BF16
    raise PcrlError(f"unsupported activation dtype {dt} (float32 or bfloat16)")
