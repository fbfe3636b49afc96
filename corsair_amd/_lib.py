"""ctypes binding of libcorsair_hip.so (the C ABI declared in include/corsair_hip.h).

The product path has no CPU fallback: if the library is missing, or a compute entry point is
called without a HIP device, this module raises.  PyTorch is used only as the owner of device
memory and streams (``tensor.data_ptr()``, ``torch.cuda.current_stream()``).
"""
from __future__ import annotations

import ctypes
import threading
import os
import re
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CORSAIR_HIP_LIB points diagnostics at an alternative build of the same library
LIB_PATH = os.environ.get("CORSAIR_HIP_LIB") or os.path.join(_HERE, "csrc", "libcorsair_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "corsair_hip.h")

_lib = None


class CorsairHipError(RuntimeError):
    """Raised when a cs_* entry point returns a negative status (message = cs_last_error())."""


def _declare(lib):
    vp = c_void_p
    sigs = {
        "cs_last_error": (c_char_p, []),
        "cs_version": (c_int, []),
        "cs_device_count": (c_int, []),
        "cs_coordmap_create": (c_int, [vp, c_int64, c_int, vp, POINTER(vp)]),
        "cs_coordmap_stride": (c_int, [vp, c_int, vp, POINTER(vp)]),
        "cs_coordmap_pyramid": (c_int, [vp, c_int64, c_int, c_int, c_int, vp, POINTER(vp)]),
        "cs_coordmap_size": (c_int64, [vp]),
        "cs_coordmap_tensor_stride": (c_int, [vp]),
        "cs_coordmap_coords": (vp, [vp]),
        "cs_coordmap_free": (None, [vp]),
        "cs_kernelmap_build": (c_int, [vp, vp, c_int, c_int, vp, POINTER(vp)]),
        "cs_kernelmap_build_many": (c_int, [c_int, POINTER(vp), POINTER(vp), POINTER(c_int), POINTER(c_int), vp, POINTER(vp)]),
        "cs_kernelmap_num_pairs": (c_int64, [vp]),
        "cs_kernelmap_rows": (c_int64, [vp]),
        "cs_kernelmap_table": (vp, [vp]),
        "cs_kernelmap_export": (c_int64, [vp, vp, vp, vp, c_int64, vp]),
        "cs_kernelmap_free": (None, [vp]),
        "cs_conv_fwd": (c_int, [vp, c_int64, c_int64, vp, c_int, c_int, vp, c_int, vp, vp, vp,
                                c_int, c_int, vp, c_int, vp]),
        "cs_conv_split_reset": (c_int, []),
        "cs_affine_act": (c_int, [c_int64, c_int, vp, c_int, vp, vp, vp, c_int, c_int, vp, c_int, vp]),
        "cs_row_l2_normalize": (c_int, [c_int64, c_int, vp, c_int, c_float, vp, c_int, vp]),
        "cs_segmented_max": (c_int, [c_int64, c_int, vp, c_int, vp, c_int, c_int, vp, vp]),
        "cs_instance_norm": (c_int, [c_int64, c_int, vp, c_int, vp, c_int, vp, vp, c_float, vp, c_int, vp]),
        "cs_voxelize": (c_int, [vp, POINTER(c_int64), c_int, c_double, vp, vp, POINTER(c_int64), vp]),
        "cs_voxelize_f64": (c_int, [vp, POINTER(c_int64), c_int, c_double, vp, vp, POINTER(c_int64), vp]),
        "cs_l2_topk": (c_int, [vp, c_int64, vp, c_int64, c_int, c_int, vp, vp, vp]),
        "cs_l2_topk_sq": (c_int, [vp, c_int64, vp, c_int64, c_int, c_int, vp, vp, vp]),
        "cs_knn_feat": (c_int, [vp, POINTER(c_int64), vp, POINTER(c_int64), POINTER(c_int32),
                                POINTER(c_int32), c_int, c_int, c_int, vp, vp, vp, vp, vp, vp]),
        "cs_chamfer_1dir": (c_int, [vp, POINTER(c_int64), vp, POINTER(c_int64), POINTER(c_int32),
                                    POINTER(c_int32), c_int, vp, vp, vp]),
        "cs_hausdorff_1dir": (c_int, [vp, POINTER(c_int64), vp, POINTER(c_int64), POINTER(c_int32),
                                      POINTER(c_int32), c_int, vp, vp, vp]),
        "cs_ransac_batch": (c_int, [vp, vp, POINTER(c_int64), c_int, c_double, c_int, c_int, c_double,
                                    c_uint64, vp, vp, vp, vp, vp]),
        "cs_ransac_prefilter_stats": (None, [POINTER(c_uint64), c_int]),
        "cs_knn_shortlist_stats": (None, [POINTER(c_uint64), c_int]),
        "cs_chamfer_f16_stats": (None, [POINTER(c_uint64), c_int]),
        "cs_l2_topk_stats": (None, [POINTER(c_uint64), c_int]),
        "cs_topk_catalog_create": (c_int, [vp, c_int64, c_int, vp, POINTER(vp)]),
        "cs_l2_topk_catalog": (c_int, [vp, c_int64, vp, c_int, vp, vp, c_int, vp]),
        "cs_topk_catalog_free": (None, [vp]),
        "cs_symcut_fit": (c_int, [vp, c_int, vp, POINTER(c_int64), c_int, vp, c_int,
                                  POINTER(c_int32), c_int, c_int, c_int, vp, vp, vp, vp, vp]),
        "cs_symcut_labels": (c_int, [vp, POINTER(c_int64), c_int, POINTER(c_int32), vp, vp, vp]),
        "cs_partition_by_label": (c_int, [vp, vp, c_int, vp, vp]),
        "cs_cfg_bad": (c_int, [vp, c_int, vp, c_int, vp, vp]),
        "cs_corr_assemble": (c_int, [vp, vp, vp, vp, c_int, vp, c_int, c_int64, vp, vp, vp]),
        "cs_prof_enable": (None, [c_int]),
        "cs_prof_reset": (None, []),
        "cs_prof_get": (c_int, [c_char_p, POINTER(c_double), POINTER(c_int64)]),
        "cs_prof_get_units": (c_int, [c_char_p, POINTER(c_double)]),
        "cs_pool_trim": (None, []),
        "cs_pool_stats": (None, [POINTER(c_uint64)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sigs


def header_symbols():
    """Names of every function include/corsair_hip.h declares (used by the CPU-side ABI test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load libcorsair_hip.so; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise CorsairHipError(load().cs_last_error().decode("utf-8", "replace"))
    return rc


def require_gpu():
    if load().cs_device_count() < 1:
        raise CorsairHipError("no HIP device visible: the corsair_amd hot path has no CPU fallback")


# ---- small helpers used by the host modules -------------------------------------------------
def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return c_void_p(torch.cuda.current_stream().cuda_stream)


def i64_array(values):
    arr = (c_int64 * len(values))(*[int(v) for v in values])
    return arr


def i32_array(values):
    arr = (c_int32 * len(values))(*[int(v) for v in values])
    return arr


def prof_enable(on=True):
    load().cs_prof_enable(1 if on else 0)


def prof_reset():
    load().cs_prof_reset()


def prof_get(name):
    ms = c_double(0.0)
    n = c_int64(0)
    check(load().cs_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n)))
    units = c_double(0.0)
    check(load().cs_prof_get_units(name.encode(), ctypes.byref(units)))
    return ms.value, n.value, units.value


# ---- device -> host downloads ---------------------------------------------------------------
_pinned = threading.local()


def to_host(*tensors):
    """NumPy copies of device tensors through page-locked staging buffers: all copies are enqueued on
    the current stream and ONE event wait follows.  `tensor.cpu()` goes through pageable memory, where
    the runtime first blocks on the stream with a slow wake-up (measured: 200-400 us of GPU idle per
    call after a long kernel) and then copies each tensor separately."""
    import torch

    cache = getattr(_pinned, "bufs", None)
    if cache is None:
        cache = _pinned.bufs = {}
    staged = []
    for i, t in enumerate(tensors):
        t = t.detach()
        key = (i, tuple(t.shape), t.dtype)
        buf = cache.get(key)
        if buf is None:
            if len(cache) > 64:
                cache.clear()
            buf = cache[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        staged.append(buf)
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()
    return [b.numpy().copy() for b in staged]
