"""ctypes binding of libeditor_hip.so.  include/editor_hip.h is the single source of truth: the
prototypes are parsed from it, so every declared symbol must exist in the library (checked at load).

There is NO fallback: if the library is missing or a symbol is absent this raises, and every op in
editor_amd.ops raises when handed a non-GPU tensor.
"""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "editor_hip.h")
LIB_PATH = os.path.join(HERE, "libeditor_hip.so")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
       "unsigned": ctypes.c_ulonglong,            # the header's only unsigned scalar type is `unsigned long long`
       "editor_stream_t": ctypes.c_void_p}


def parse_header(path=HEADER):
    """-> {name: [ctypes argtypes]} for every `int editor_*(...)` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(editor_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(2).split(","):
            a = a.strip()
            if "*" in a:
                args.append(ctypes.c_void_p)
            else:
                ty = a.replace("const", "").split()[0]
                args.append(_CT[ty])
        protos[m.group(1)] = args
    return protos


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda d: torch.cuda.current_stream(d).cuda_stream)


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libeditor_hip.so not built - run `python -m editor_amd.build` (hipcc, gfx950). "
                "There is no CPU/PyTorch fallback for the EDITOR hot path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self._fn = {}
        self.protos = parse_header()
        for name, argtypes in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the header promises too much
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int

    def call(self, name, *args):
        fn = self._fn.get(name)
        if fn is None:
            fn = self._fn[name] = getattr(self.cdll, name)
        conv = []
        dev = -1
        for a in args:
            if isinstance(a, torch.Tensor):
                if not a.is_cuda:
                    raise RuntimeError(f"{name}: tensor argument is not on the GPU (no CPU fallback)")
                if not a.is_contiguous():
                    raise RuntimeError(f"{name}: non-contiguous tensor")
                if dev < 0:
                    dev = a.device.index
                conv.append(a.data_ptr())
            else:
                conv.append(a)                       # None -> NULL; ints / floats / ctypes arrays as they are
        # raw handle of the current stream of the arguments' device (the Stream-object route costs ~2 us per launch)
        cur = torch.cuda.current_device()
        if dev < 0:
            dev = cur
        conv.append(_raw_stream(dev))
        if dev != cur:                       # model on cuda:N without set_device(N): launch on the tensors' device
            with torch.cuda.device(dev):
                rc = fn(*conv)
        else:
            rc = fn(*conv)
        if rc != 0:
            raise RuntimeError(f"{name} failed: hipError {rc}")


_LIB = None
_PROBE = None


def probe_lib():
    """libeditor_probe.so (include/editor_debug.h): hardware-semantics probes for tests; never on the product path."""
    global _PROBE
    if _PROBE is None:
        _PROBE = ctypes.CDLL(os.path.join(HERE, "libeditor_probe.so"))
    return _PROBE


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def call(name, *args):
    lib().call(name, *args)
