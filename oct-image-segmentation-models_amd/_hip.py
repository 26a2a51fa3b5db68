"""ctypes binding of ``liboct_unet_hip.so`` (C ABI declared in ``include/oct_unet.h``).

The library is built in-tree by ``csrc/build.sh`` (``__graft_entry__.build()``).
Importing this module without the built library raises: the product path has
no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCT_UNET_LIB") or os.path.join(_HERE, "liboct_unet_hip.so")     # (OCT_UNET_LIB: kernel experiments)


class OctError(RuntimeError):
    pass


class UNetCfg(C.Structure):
    _fields_ = [
        ("in_ch", C.c_int), ("n_cls", C.c_int), ("H", C.c_int), ("W", C.c_int), ("max_batch", C.c_int),
        ("start_neurons", C.c_int), ("pool_layers", C.c_int), ("conv_layers", C.c_int),
        ("enc_k", C.c_int), ("dec_k", C.c_int), ("dtype", C.c_int), ("training", C.c_int),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("dropout_rate", C.c_float),
        ("bn_unbiased_moving_var", C.c_int), ("seed", C.c_ulonglong),
    ]


class LayerInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 32),
        ("kh", C.c_int), ("kw", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("has_bn", C.c_int),
        ("out_h", C.c_int), ("out_w", C.c_int),
        ("kernel_off", C.c_size_t), ("bias_off", C.c_size_t), ("gamma_off", C.c_size_t), ("beta_off", C.c_size_t),
        ("moving_mean_off", C.c_size_t), ("moving_var_off", C.c_size_t),
    ]


class ProfileEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 80), ("layer", C.c_char * 32), ("launches", C.c_int),
                ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class UNetIO(C.Structure):
    _fields_ = [("probs", C.c_void_p), ("argmax", C.c_void_p), ("labels", C.c_void_p)]


# every symbol include/oct_unet.h declares: (name, restype, argtypes)
_P = C.POINTER
SYMBOLS = [
    ("oct_unet_cfg_default", None, [_P(UNetCfg)]),
    ("oct_unet_cfg_check", C.c_int, [_P(UNetCfg)]),
    ("oct_unet_param_count", C.c_size_t, [_P(UNetCfg)]),
    ("oct_unet_state_count", C.c_size_t, [_P(UNetCfg)]),
    ("oct_unet_workspace_bytes", C.c_size_t, [_P(UNetCfg)]),
    ("oct_unet_layer_count", C.c_int, [_P(UNetCfg)]),
    ("oct_unet_layer_info", C.c_int, [_P(UNetCfg), C.c_int, _P(LayerInfo)]),
    ("oct_unet_create", C.c_int, [_P(UNetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  _P(C.c_void_p)]),
    ("oct_unet_destroy", None, [C.c_void_p]),
    ("oct_unet_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _P(UNetIO), C.c_void_p]),
    ("oct_unet_loss_dice", C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    ("oct_unet_set_focal_dice", C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    ("oct_unet_loss_focal_dice", C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    ("oct_unet_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    ("oct_unet_set_tail_event", C.c_int, [C.c_void_p, C.c_void_p]),
    ("oct_unet_grad_tail_offset", C.c_size_t, [_P(UNetCfg)]),
    ("oct_adam_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_long, C.c_void_p]),
    ("oct_sgd_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]),
    ("oct_unet_set_dropout_step", C.c_int, [C.c_void_p, C.c_ulonglong]),
    ("oct_unet_dropout_mask", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("oct_unet_graph_capture", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _P(UNetIO), C.c_void_p]),
    ("oct_unet_graph_launch", C.c_int, [C.c_void_p, C.c_void_p]),
    ("oct_unet_profile_begin", C.c_int, [C.c_void_p]),
    ("oct_unet_profile_end", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _P(C.c_int)]),
    ("oct_boundary_maps", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("oct_set_option", C.c_int, [C.c_char_p, C.c_int]),
    ("oct_get_option", C.c_int, [C.c_char_p, _P(C.c_int)]),
    ("oct_unet_get_option", C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_int)]),
    ("oct_unet_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("oct_unet_debug_activation", C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    ("oct_unet_debug_layer_fused", C.c_int, [C.c_void_p, C.c_int]),
    ("oct_last_error", C.c_char_p, []),
    ("oct_version", C.c_char_p, []),
]

_lib = None


def source_stamp() -> str:
    """SHA-256 (12 hex digits) of csrc/*.hip, csrc/*.hpp and include/*.h in build.sh's order, or '' when the sources
    are not beside the package (an installed copy)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc, inc = os.path.join(here, "csrc"), os.path.join(here, "..", "include")
    files = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(inc, "*.h"))
    if not files:
        return ""
    # build.sh sorts the names as it lists them: '*.hip *.hpp ../../include/*.h' through `sort`
    rel = sorted((os.path.basename(f) if os.path.dirname(f) == csrc else "../../include/" + os.path.basename(f), f) for f in files)
    h = hashlib.sha256()
    for _, f in rel:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OctError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(csrc/build.sh).  There is no fallback implementation.")
        # torch ships its own HIP runtime (libamdhip64): it must be in the process BEFORE this library is loaded, so
        # that both resolve to the same runtime instance (otherwise: "no ROCm-capable device is detected")
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        # a stale binary (built from other sources than the ones beside it) must not pass for this tree
        want, have = source_stamp(), l.oct_version().decode()
        if want and f"src:{want}" not in have and not os.environ.get("OCT_ALLOW_STALE_LIB"):
            raise OctError(f"{LIB_PATH} is stale: it reports '{have}' but the sources beside it hash to src:{want}; "
                           "rebuild with csrc/build.sh (OCT_ALLOW_STALE_LIB=1 overrides)")
        _lib = l
        # tuning knobs for experiments: OCT_OPTIONS="name=value,name=value" (oct_set_option; results do not depend on them)
        for kv in filter(None, os.environ.get("OCT_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            if l.oct_set_option(k.strip().encode(), int(v)) != 0:
                raise OctError(f"OCT_OPTIONS: {l.oct_last_error().decode()}")
    return _lib


def set_option(name: str, value: int) -> None:
    check(lib().oct_set_option(name.encode(), int(value)), f"oct_set_option({name})")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib().oct_get_option(name.encode(), C.byref(v)), f"oct_get_option({name})")
    return int(v.value)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise OctError(f"{what} failed (rc={rc}): {lib().oct_last_error().decode()}")
