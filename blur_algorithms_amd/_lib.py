"""ctypes binding of the C ABI (include/blur_amd.h) in libblur_amd.so.

There is no fallback: if the HIP library is missing this module raises, and every call
that needs the GPU returns the library's error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLUR_AMD_LIB") or os.path.join(_HERE, "libblur_amd.so")   # BLUR_AMD_LIB: developer A/B builds

BLUR_OK = 0
ERR_NAMES = {1: "BLUR_ERR_INVALID", 2: "BLUR_ERR_UNSUPPORTED", 3: "BLUR_ERR_HIP", 4: "BLUR_ERR_NOMEM"}


class BlurError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "BLUR_ERR"), code, msg))
        self.code = code


class BlurOpts(C.Structure):
    _fields_ = [("nyquist_quirk", C.c_int), ("col_group", C.c_int), ("force_generic", C.c_int), ("frames_per_launch", C.c_int),
                ("row_major_planes", C.c_int), ("engine", C.c_int), ("tile_points", C.c_int), ("reserved", C.c_int * 1)]


# every symbol include/blur_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "blur_opts_default": (None, [C.POINTER(BlurOpts)]),
    "blur_gaussian_window": (C.c_int, [C.c_double, C.c_int]),
    "blur_get_gaussian": (C.c_int, [_P, C.c_double, C.c_int, C.c_int]),
    "blur_is_valid_size": (C.c_int, [C.c_int]),
    "blur_nearest_transform_size": (C.c_int, [C.c_int]),
    "blur_pffft_sizing": (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]),
    "blur_kernel_multipliers": (C.c_int, [C.c_double, C.c_int, C.c_int, _P]),
    "blur_fft_plan_radices": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "blur_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "blur_ctx_destroy": (C.c_int, [_P]),
    "blur_ctx_set_stream": (C.c_int, [_P, _P]),
    "blur_ctx_synchronize": (C.c_int, [_P]),
    "blur_last_error": (C.c_char_p, [_P]),
    "blur_last_engine": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "blur_ctx_timing_enable": (C.c_int, [_P, C.c_int]),
    "blur_ctx_timing": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "blur_gaussian_u8c3_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_u8c3_batch_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_f32c1_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_u8c3_host": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_u8c3_host_batch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_u8c3_host_pitched": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_f32c1_host": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_separable_u8c3_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.POINTER(BlurOpts)]),
    "blur_boxfft_u8c3_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_boxfft_sizing": (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]),
    "blur_box_kernel": (C.c_int, [_P, C.c_int, C.c_int]),
    "blur_rowpass_u8c3_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_pocketfft2d_sizing": (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int * 8)]),
    "blur_pocketfft2d_u8c3_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_int, _P]),
    "blur_pocketfft2d_u8c3_host": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_int]),
    "blur_reflect101_u8_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int * 2)]),
    "blur_flip_block_f32_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "blur_deinterleave_bgr_u8_f32_dev": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "blur_interleave_bgr_f32_u8_dev": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "blur_fastboxblur_u8_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "blur_fastboxblur_u8_host": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "blur_multi_create": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.c_int]),
    "blur_multi_destroy": (C.c_int, [_P]),
    "blur_multi_shards": (C.c_int, [_P]),
    "blur_multi_last_error": (C.c_char_p, [_P]),
    "blur_gaussian_u8c3_batch_multi_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_gaussian_u8c3_batch_multi_host": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(BlurOpts)]),
    "blur_convolve_lines_c32_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "blur_wr_length": (C.c_int, [C.c_int, C.c_int]),
    "blur_mx_window_blocks": (C.c_int, [C.c_int]),
    "blur_mx_fragments": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "blur_wr_kernel_multipliers": (C.c_int, [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "blur_malloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "blur_free": (C.c_int, [_P, _P]),
    "blur_host_alloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "blur_host_free": (C.c_int, [_P, _P]),
    "blur_copy_bandwidth": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "blur_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "blur_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
}

_lib = None


def load():
    """Load libblur_amd.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "blur_algorithms_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C blur_algorithms_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        # share the HIP runtime torch has already loaded (same SONAME) when torch is in use
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the library
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
