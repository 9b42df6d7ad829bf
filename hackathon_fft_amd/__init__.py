"""mifft -- MI355X-native batched N-D radix-N FFT (drop-in for the GPU hot path of
martinvuyk/hackathon-fft).  See include/mifft.h for the C ABI, DESIGN.md for the kernels.
"""
from ._lib import MifftError, LIB_PATH, EXPORTS  # noqa: F401
from .api import (  # noqa: F401
    DeviceContext,
    GPUTest,
    Plan,
    clear_plan_cache,
    estimate_best_bases,
    estimate_best_bases_nd,
    fft,
    fftn,
    ifftn,
    ordered_bases,
    plan_fft,
    rfftn,
    time_fft,
)

__all__ = [
    "DeviceContext", "GPUTest", "Plan", "clear_plan_cache", "MifftError", "estimate_best_bases", "estimate_best_bases_nd",
    "fft", "fftn", "ifftn", "ordered_bases", "plan_fft", "rfftn", "time_fft",
]
