"""ctypes binding of libryolo_hip.so (include/ryolo.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every kernel on the hot path lives in
csrc/*.hip and is reached through the C ABI with raw device pointers.  There is NO fallback: if the shared library
is missing or a tensor is not on a HIP device the call raises (the product path must fail loudly, never silently
run something else).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RYOLO_LIB", os.path.join(_HERE, "csrc", "libryolo_hip.so"))     # RYOLO_LIB: A/B builds of the same ABI
_ERR = {1: "invalid argument", 2: "workspace too small", 3: "kernel launch failed", 4: "unsupported size/configuration"}

_P, _I, _L, _F, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
_SIGNATURES = {
    "ryolo_nms_workspace_bytes": [_I, _L, ctypes.POINTER(_Z)],
    "ryolo_nms_rotated_batched": [_P, _P, _I, _L, _F, _I, _L, _P, _Z, _P, _L, _P, _P],
    "ryolo_box_iou_rotated": [_P, _I, _P, _I, _P, _Z, _P, _P],
    "ryolo_diag_iou_rotated": [_P, _P, _I, _P, _P],
    "ryolo_head_permute": [_P, _P, _I, _I, _I, _I, _P],
    "ryolo_decode": [_I, _P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(_F), _L, _L, _P],
    "ryolo_pp_score": [_P, _I, _L, _I, _F, _P, _P, _P, _P],
    "ryolo_pp_gather": [_P, _P, _P, _P, _I, _L, _I, _L, _L, _F, _P, _P, _P, _P],
    "ryolo_sort_workspace_bytes": [_I, _L, ctypes.POINTER(_Z)],
    "ryolo_topk_desc": [_P, _I, _L, _I, _P, _P, _P, _P, _Z, _P],
    "ryolo_argsort_desc": [_P, _L, _P, _P, _Z, _P],
    "ryolo_paste_rects": [_P, _P, _I, _P, _I, _I, _I, _I, _P],
    "ryolo_paste_rects_grouped": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "ryolo_paste_rect_bytes": [ctypes.POINTER(_I)],
    "ryolo_warp_perspective_u8": [_P, _I, _I, _I, _P, _P, _I, _I, _I, _P],
    "ryolo_hsv_gain_u8": [_P, _L, _P, _P],
    "ryolo_mixup_u8": [_P, _P, ctypes.c_double, _L, _P, _P],
    "ryolo_letterbox_u8": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "ryolo_resize_item_bytes": [ctypes.POINTER(_I)],
    "ryolo_resize_hsv_batch": [_P, _P, _I, _L, _P, _P, _P],
    "ryolo_label_row_bytes": [ctypes.POINTER(_I)],
    "ryolo_label_stage": [_P, _L, _P, _P, _P],
    "ryolo_pp_emit": [_P, _P, _P, _I, _L, _L, _P, _P],
    "ryolo_map_match_workspace_bytes": [_L, _L, ctypes.POINTER(_Z)],
    "ryolo_map_match": [_P, _P, _P, _P, _I, _L, _L, _P, _I, _I, _P, _P, _Z, _P],
    "ryolo_ap_workspace_bytes": [_L, _I, _I, ctypes.POINTER(_Z)],
    "ryolo_ap_per_class": [_P, _P, _P, _L, _P, _L, _I, _I, _P, _P, _I, _P, _Z, _P, _P, _P, _P, _P, _P],
    "ryolo_to_tensor": [_P, _I, _I, _I, _P, _P, _P],
    "ryolo_encode_labels": [_P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P],
    "ryolo_polys_to_xywha": [_P, _L, _P, _P],
    "ryolo_dets_to_polys": [_P, _P, _P, _I, _I, _L, _P, _P],
}
_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into csrc/libryolo_hip.so (hipcc cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "libryolo_hip.so", "-j8"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)


def register(name, argtypes):
    _SIGNATURES[name] = argtypes
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"ryolov4_amd: HIP library not built ({LIB_PATH}). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no CPU/PyTorch fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        from .engine import structs  # noqa: F401  (registers the conv-stack / loss entry points)
        for name, sig in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError => header/library mismatch, fail loudly
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        _lib = L
    return _lib


def exported_symbols():
    from .engine import structs  # noqa: F401  (registers the conv-stack / loss entry points)
    return sorted(_SIGNATURES)


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("ryolov4_amd: tensor is not on a HIP device (no CPU fallback on the product path)")
    if not t.is_contiguous():
        raise RuntimeError("ryolov4_amd: tensor must be contiguous")
    return t.data_ptr()


def require_device(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"ryolov4_amd.{what}: expected a tensor on a HIP device (MI355X); the product path has no CPU "
                           "fallback — use oracle/ for CPU reference results")
