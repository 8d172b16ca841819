"""blur_algorithms_amd -- MI355X (gfx950) engine for the FFT Gaussian-blur hot path of
michelerenzullo/Blur_algorithms (pffft_(), Source.cpp:429-570, and fastboxblur).

The product is libblur_amd.so (hand-written HIP + a C ABI, include/blur_amd.h); this package
is the thin host-side mirror of the reference's call surface over it.
"""
from ._lib import BlurError, LIB_PATH  # noqa: F401
from .api import (BlurContext, BlurMulti, gaussian_window, getGaussian, isValidSize, nearestTransformSize,  # noqa: F401
                  pffft_sizing, kernel_multipliers, fft_plan_radices, box_kernel, boxfft_sizing)
