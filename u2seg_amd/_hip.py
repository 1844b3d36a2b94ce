"""ctypes binding of libu2seg_hip.so (the C ABI declared in include/u2seg_hip.h).

There is deliberately no fallback: if the library is missing or a launcher reports an error the
caller gets an exception.  The binding style mirrors how the reference reaches its native ops
(detectron2/layers/roi_align_rotated.py:9-46 calling torch.ops.detectron2.* registered in
detectron2/layers/csrc/vision.cpp:111-116): thin Python -> C launcher -> device kernel.
"""
import ctypes
import os
import re

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libu2seg_hip.so")
_HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "u2seg_hip.h")

_lib = None
_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
}


class HipLibraryMissing(RuntimeError):
    pass


def declared_symbols(header_path=_HEADER):
    """Parse include/u2seg_hip.h -> {name: (restype, [argtype, ...])}.  Any pointer is a c_void_p."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|long long)\s+(u2_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = "long long" if a.startswith("long long") else a.split()[0]
                    if a.startswith("signed char") or a.startswith("unsigned"):
                        raise ValueError("unsupported scalar arg in header: " + a)
                    argtypes.append(_CTYPES[base])
        out[name] = (_CTYPES[ret], argtypes)
    return out


def lib_path():
    return _LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryMissing(
            "libu2seg_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(u2seg_amd/csrc/build.sh). There is no CPU fallback for the hot path." % _LIB_PATH
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in declared_symbols().items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _conv(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


# the raw handle of torch's current stream without building a torch.cuda.Stream object per launch: under cProfile the object form is
# 40 % of a binding call (tools/exp/host_profile.py), and the ROI heads' forward phase is host-bound
_raw_stream = None if os.environ.get("U2_HIP_STREAM_OBJECT", "0") == "1" else getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Launch `name` on torch's current stream (appended as the last argument)."""
    lib = load()
    rc = getattr(lib, name)(*[_conv(a) for a in args], stream_ptr())
    if rc != 0:
        raise RuntimeError("%s failed with status %d" % (name, rc))


def call_status(name, *args):
    """Like `call`, for launchers that answer 1 = "shape not served, take the other path": returns that status (0 or 1) and raises
    on anything else."""
    lib = load()
    rc = getattr(lib, name)(*[_conv(a) for a in args], stream_ptr())
    if rc not in (0, 1):
        raise RuntimeError("%s failed with status %d" % (name, rc))
    return rc


def call_nostream(name, *args):
    lib = load()
    return getattr(lib, name)(*[_conv(a) for a in args])
