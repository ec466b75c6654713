"""ctypes binding of libpeanut_hip.so (the C ABI declared in include/peanut_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, every product
entry point raises.  ``import torch`` happens before the dlopen so that the library binds to the
HIP runtime torch already loaded (same ``libamdhip64.so.7`` soname -> one runtime per process,
torch device pointers and streams are directly usable).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

from . import build as _build

_LOCK = threading.Lock()
_LIB = None
ABI_VERSION = 15    # must equal peanut_abi_version() of the loaded library (struct layouts, argument lists)


class PeanutHipError(RuntimeError):
    pass


class PeanutRangeError(PeanutHipError, FloatingPointError):
    """PEANUT_ERANGE: precision fp16x3 and a value left fp16's exponent range (include/peanut_hip.h)."""


class PredCfgC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("num_classes", C.c_int),
        ("strides", C.c_int * 4), ("dilations", C.c_int * 4), ("contract_dilation", C.c_int),
        ("pool_scales", C.c_int * 8), ("n_pool_scales", C.c_int),
        ("head_channels", C.c_int), ("align_corners", C.c_int), ("bn_eps", C.c_float),
        ("precision", C.c_int), ("fold_ppm", C.c_int), ("conv_algo", C.c_int),
    ]


class MapCfgC(C.Structure):
    _fields_ = [
        ("frame_height", C.c_int), ("frame_width", C.c_int), ("map_resolution", C.c_int),
        ("map_size_cm", C.c_int), ("global_downscaling", C.c_int), ("vision_range", C.c_int),
        ("hfov", C.c_double), ("du_scale", C.c_int),
        ("cat_pred_threshold", C.c_double), ("exp_pred_threshold", C.c_double),
        ("map_pred_threshold", C.c_double), ("num_sem_categories", C.c_int), ("camera_height", C.c_double),
    ]


class RcnnCfgC(C.Structure):
    _fields_ = [
        ("depth", C.c_int), ("stem_out", C.c_int), ("res2_out", C.c_int), ("stride_in_1x1", C.c_int),
        ("fpn_out", C.c_int), ("num_anchors", C.c_int), ("min_size", C.c_int), ("max_size", C.c_int),
        ("size_divisibility", C.c_int), ("pixel_mean", C.c_float * 3), ("pixel_std", C.c_float * 3),
        ("bn_eps", C.c_float), ("precision", C.c_int), ("conv_algo", C.c_int),
        ("anchor_sizes", C.c_float * 5), ("aspect_ratios", C.c_float * 8),
        ("rpn_pre_nms_topk", C.c_int), ("rpn_post_nms_topk", C.c_int), ("rpn_nms_thresh", C.c_float),
        ("rpn_bbox_weights", C.c_float * 4), ("num_classes", C.c_int), ("box_pooler_resolution", C.c_int),
        ("mask_pooler_resolution", C.c_int), ("fc_dim", C.c_int), ("mask_conv_dim", C.c_int), ("num_mask_convs", C.c_int),
        ("roi_bbox_weights", C.c_float * 4), ("score_thresh_test", C.c_float), ("nms_thresh_test", C.c_float),
        ("detections_per_image", C.c_int), ("mask_threshold", C.c_float),
    ]


class TensorC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


# every symbol include/peanut_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "peanut_last_error": (C.c_char_p, []),
    "peanut_last_conv_kernel": (C.c_char_p, []),
    "peanut_abi_version": (C.c_int, []),
    "peanut_build_arch": (C.c_char_p, []),
    "peanut_source_hash": (C.c_char_p, []),
    "peanut_debug_weight_pieces": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "peanut_debug_wino_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "peanut_debug_deferred_splitk_count": (C.c_longlong, []),
    "peanut_debug_lds_canary": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "peanut_debug_pkfma_canary": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "peanut_pred_create": (C.c_int, [C.POINTER(_P), C.POINTER(PredCfgC), C.POINTER(TensorC), C.c_int]),
    "peanut_pred_destroy": (None, [_P]),
    "peanut_pred_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "peanut_pred_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    "peanut_pred_flops_per_map": (C.c_double, [_P, C.c_int, C.c_int]),
    "peanut_pred_debug_keep": (C.c_int, [_P, C.c_int]),
    "peanut_pred_debug_tensor": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int * 4)]),
    "peanut_pred_debug_read": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_int * 4), _P]),
    "peanut_pred_probe_enable": (C.c_int, [_P, C.c_int]),
    "peanut_pred_use_graph": (C.c_int, [_P, C.c_int]),
    "peanut_map_use_graph": (C.c_int, [_P, C.c_int]),
    "peanut_map_mark_agent": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "peanut_pred_probe_collect": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                            C.POINTER(C.c_int)]),
    "peanut_map_create": (C.c_int, [C.POINTER(_P), C.POINTER(MapCfgC)]),
    "peanut_map_destroy": (None, [_P]),
    "peanut_map_dims": (C.c_int, [_P, C.POINTER(C.c_int * 4)]),
    "peanut_map_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "peanut_rcnn_create": (C.c_int, [C.POINTER(_P), C.POINTER(RcnnCfgC), C.POINTER(TensorC), C.c_int]),
    "peanut_rcnn_destroy": (None, [_P]),
    "peanut_rcnn_plan": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int * 2), C.POINTER(C.c_int * 2),
                                   C.POINTER(C.c_int * 10), C.POINTER(C.c_size_t), C.POINTER(C.c_double)]),
    "peanut_rcnn_forward_front": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(_P),
                                            C.POINTER(_P), _P]),
    "peanut_rcnn_probe_front": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "peanut_rcnn_set_stage_timing": (C.c_int, [_P, C.c_int]),
    "peanut_rcnn_stage_times": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "peanut_rcnn_preprocess": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "peanut_rcnn_inference": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _P, _P, _P, _P, _P]),
    "peanut_rcnn_semantic": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int32), _P,
                                       C.POINTER(C.c_int), _P, _P, _P, _P, _P]),
    "peanut_rcnn_debug_stage": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "peanut_roi_align": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_int, _P, _P,
                                   C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "peanut_nms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "peanut_nms": (C.c_int, [_P, _P, C.c_int, C.c_float, _P, _P, _P]),
    "peanut_nms_segments": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.c_int, C.c_float, _P, _P, _P]),
    "peanut_paste_masks": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "peanut_preprocess_obs": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P]),
    "peanut_seg_accumulate": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_int, _P, _P]),
    "peanut_goal_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int]),
    "peanut_goal_destroy": (None, [_P]),
    "peanut_goal_reset": (C.c_int, [_P]),
    "peanut_goal_rounds": (C.c_int, [_P]),
    "peanut_goal_passes": (C.c_int, [_P]),
    "peanut_goal_select_begin": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "peanut_goal_converged": (C.c_int, [_P]),
    "peanut_goal_traversible": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "peanut_fmm_distance": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "peanut_goal_select": (C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_int * 4), C.c_int, C.c_int, _P, C.c_double, C.c_int,
                                     C.POINTER(C.c_int * 2), C.POINTER(C.c_double * 4), _P, _P, _P]),
    "peanut_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte * 128)]),
    "peanut_comm_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.POINTER(C.c_ubyte * 128)]),
    "peanut_comm_destroy": (None, [_P]),
    "peanut_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "peanut_comm_backend": (C.c_char_p, []),
    "peanut_allgather_maps": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "peanut_conv_create": (C.c_int, [C.POINTER(_P), _P, _P, _P] + [C.c_int] * 11),
    "peanut_conv_destroy": (None, [_P]),
    "peanut_conv_precision": (C.c_int, [_P]),
    "peanut_conv_forward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "peanut_set_default_option": (C.c_int, [C.c_char_p, C.c_longlong]),
    "peanut_get_default_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "peanut_option_list": (C.c_char_p, []),
    "peanut_pred_set_option": (C.c_int, [_P, C.c_char_p, C.c_longlong]),
    "peanut_pred_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_longlong)]),
    "peanut_conv_set_option": (C.c_int, [_P, C.c_char_p, C.c_longlong]),
    "peanut_rcnn_set_option": (C.c_int, [_P, C.c_char_p, C.c_longlong]),
}


CONV_ALGOS = {"auto": 0, "direct": 1}   # PEANUT_ALGO_*
PRECISIONS = {"fp32": 0, "bf16x3": 1, "fp16x3": 2, "bf16x6": 3}   # PEANUT_PREC_*


def lib_path() -> str:
    return os.environ.get("PEANUT_HIP_LIB", _build.LIB_PATH)


def _stale_reason(lib, path) -> str:
    """'' when the loaded in-tree library was built from the csrc/ + include/ sources lying next to it (content hash,
    not file times), else what differs.  A library given through PEANUT_HIP_LIB is taken as it is."""
    if path != _build.LIB_PATH or not os.path.isdir(_build.CSRC):
        return ""
    have = (lib.peanut_source_hash() or b"").decode()
    want = _build.source_hash()
    return "" if have == want else f"built from sources {have or '<unknown>'}, the tree holds {want}"


def load() -> C.CDLL:
    """dlopen libpeanut_hip.so and attach signatures; raises PeanutHipError if it is absent, exports the wrong ABI, or is
    stale with respect to the sources next to it and cannot be rebuilt."""
    global _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = lib_path()
        if path == _build.LIB_PATH and os.path.exists(path) and os.path.isdir(_build.CSRC) and _build._built_hash() and \
                _build._built_hash() != _build.source_hash():
            try:                                   # sources edited since the last build: rebuild before the first dlopen
                _build.build()
            except Exception as e:  # noqa: BLE001
                raise PeanutHipError(f"{path} is stale (csrc/ changed since it was built) and rebuilding failed: {e}") from e
        if not os.path.exists(path):
            raise PeanutHipError(
                f"HIP extension not built: {path} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m peanut_amd.build`). "
                "peanut_amd has no CPU fallback.")
        try:
            lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        except OSError as e:  # pragma: no cover - environment specific
            raise PeanutHipError(f"failed to load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise PeanutHipError(f"{path} does not export {name}; rebuild the extension") from e
            fn.restype = res
            fn.argtypes = args
        got = lib.peanut_abi_version()
        if got != ABI_VERSION:
            raise PeanutHipError(f"{path} implements ABI version {got}, this package binds version {ABI_VERSION}: "
                                 "rebuild the extension (python -m peanut_amd.build --force)")
        why = _stale_reason(lib, path)
        if why:
            raise PeanutHipError(f"{path} is stale: {why}.  Rebuild it (python -m peanut_amd.build --force); a stale kernel "
                                 "library must not answer for the sources next to it")
        _LIB = lib
        return lib


_OPTION_LOCK = threading.RLock()


class default_options:
    """``with default_options(wino_m=6, pw256_mink=2048): handle = ...`` -- changes the process defaults of the library's
    tuning options (csrc/options.h; ``peanut_option_list()`` names them) for the handles CREATED inside the block and
    restores them afterwards.  A handle snapshots the defaults when it is created, so this is how the create-time options
    (Winograd forms, packing tiles) reach one handle without touching the environment; run-time options can also be changed
    on a live handle (``peanut_pred_set_option`` / ``peanut_conv_set_option`` / ``peanut_rcnn_set_option``).

    The block holds a process-wide re-entrant lock: the library's defaults are one global table without locking of its own, and a
    handle created on another thread in the middle of someone's block would inherit their values.  EVERY Python mirror therefore
    creates its handle inside a (possibly empty) ``default_options()`` block; direct C callers that create handles from several
    threads while changing defaults must serialise those calls themselves (include/peanut_hip.h)."""

    def __init__(self, **options):
        self.options = {k: int(v) for k, v in options.items()}
        self.saved = {}

    def __enter__(self):
        lib = load()
        _OPTION_LOCK.acquire()
        try:
            for k, v in self.options.items():
                old = C.c_longlong()
                check(lib.peanut_get_default_option(k.encode(), C.byref(old)), "peanut_get_default_option")
                self.saved[k] = old.value
                check(lib.peanut_set_default_option(k.encode(), v), "peanut_set_default_option")
        except Exception:
            self.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, *exc):
        lib = load()
        try:
            for k, v in self.saved.items():
                lib.peanut_set_default_option(k.encode(), v)
            self.saved = {}
        finally:
            _OPTION_LOCK.release()
        return False


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().peanut_last_error()
        if rc == -5:
            raise PeanutRangeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
        raise PeanutHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def current_stream_ptr(device=None) -> int:
    """Raw hipStream_t of torch's current stream on ``device`` (kernels are enqueued there)."""
    return int(torch.cuda.current_stream(device).cuda_stream)
