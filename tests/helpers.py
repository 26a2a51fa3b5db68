"""Shared test helpers (CPU side): numpy replica of the engine's counter-based dropout stream and a
ReLU-boundary margin check that justifies tight fp32-vs-fp64 tolerances."""
import numpy as np

from oracle import unet_numpy as on


def drop_hash(seed: int, step: int, idx: np.ndarray) -> np.ndarray:
    """Replica of oct::drop_hash (csrc/common.hpp)."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (np.uint64(step) * np.uint64(0x9E3779B97F4A7C15)) \
            ^ (idx.astype(np.uint64) * np.uint64(0xD1B54A32D192ED03))
        x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return (x >> np.uint64(32)).astype(np.uint32)


def dropout_keep_mask(seed: int, step: int, shape, rate: float = 0.5) -> np.ndarray:
    n = int(np.prod(shape))
    thresh = min(4294967295, int(np.floor(rate * 4294967296.0)))
    return (drop_hash(seed, step, np.arange(n, dtype=np.uint32)) >= np.uint32(thresh)).reshape(shape)


def relu_margin(cfg, params, cache) -> float:
    """Smallest |gamma*xhat+beta| over all BN blocks of a training forward: if it is well above the fp32
    rounding of the device path, no ReLU mask (and no pool arg-max among positive values) can flip."""
    m = np.inf
    for li, spec in enumerate(on.build_plan(cfg)):
        if spec.has_bn:
            yb = params[li]["gamma"] * cache[li]["xhat"] + params[li]["beta"]
            m = min(m, float(np.abs(yb).min()))
    return m
