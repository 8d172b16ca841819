"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so) and of the
reference-built helper library (oracle/_ref/libref_utils.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from blur_algorithms_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_utils.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is mounted)."""
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/Utils.hpp") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # BLUR_ORACLE_LIB: bench.py's timed CPU baseline loads a -march=native rebuild made on the host it runs on
        L = C.CDLL(os.environ.get("BLUR_ORACLE_LIB") or _LIB)
        L.ora_gaussian_window.argtypes = [C.c_double, C.c_int]
        L.ora_gaussian_window.restype = C.c_int
        L.ora_get_gaussian.argtypes = [_f32p, C.c_double, C.c_int, C.c_int]
        L.ora_is_valid_size.argtypes = [C.c_int]
        L.ora_nearest_transform_size.argtypes = [C.c_int]
        L.ora_pffft_sizing.argtypes = [C.c_int, C.c_int, C.c_double, _i32p]
        L.ora_deinterleave_bgr_u8_f32.argtypes = [_u8p, _f32p, _f32p, _f32p, C.c_uint32]
        L.ora_interleave_bgr_f32_u8.argtypes = [_f32p, _f32p, _f32p, _u8p, C.c_uint32]
        L.ora_reflect_101.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p]
        L.ora_flip_block_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int]
        L.ora_pad_tile.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p]
        L.ora_fft64.argtypes = [C.c_int, _f64p, _f64p, C.c_int]
        L.ora_kernel_multipliers.argtypes = [C.c_double, C.c_int, C.c_int, _f32p]
        L.ora_pffft_plane_f64.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.ora_pffft_plane_f64.restype = C.c_int
        L.ora_pffft_blur_u8c3_f64.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.ora_pffft_blur_u8c3_f64.restype = C.c_int
        L.ora_pffft_blur_u8c3_f32.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_double]
        L.ora_pffft_blur_u8c3_f32.restype = C.c_int
        L.ora_fft_new_setup.argtypes = [C.c_int]
        L.ora_fft_new_setup.restype = C.c_void_p
        L.ora_fft_destroy_setup.argtypes = [C.c_void_p]
        L.ora_fft_transform_ordered.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int]
        L.ora_sorted_optimized_convolution.argtypes = [_f32p, _f32p, C.c_int, C.c_float]
        L.ora_fastboxblur_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ora_fastboxblur_u8.restype = C.c_int
        L.ora_num_threads.restype = C.c_int
        L.ora_box_kernel_1d.argtypes = [_f32p, C.c_int, C.c_int]
        L.ora_kernel_multipliers_from_array.argtypes = [_f32p, C.c_int, _f32p]
        L.ora_pffft_blur_u8c3_f64_kernel.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_void_p]
        L.ora_pffft_blur_u8c3_f64_kernel.restype = C.c_int
        L.ora_boxfft_sizing.argtypes = [C.c_int, C.c_int, C.c_double, _i32p]
        _lib = L
    return _lib


# --------------------------------------------------------------------------
# restatement (oracle) entry points
# --------------------------------------------------------------------------
def gaussian_window(sigma, max_width=0):
    return lib().ora_gaussian_window(float(sigma), int(max_width))


def get_gaussian(sigma, width=0, fft_length=0):
    w = width or gaussian_window(sigma)
    k = np.zeros(max(w, fft_length), np.float32)
    lib().ora_get_gaussian(k, float(sigma), int(width), int(fft_length))
    return k


def is_valid_size(n):
    return lib().ora_is_valid_size(int(n))


def nearest_transform_size(n):
    return lib().ora_nearest_transform_size(int(n))


def pffft_sizing(rows, cols, sigma):
    o = np.zeros(6, np.int32)
    lib().ora_pffft_sizing(rows, cols, float(sigma), o)
    return dict(kSize=int(o[0]), pad=int(o[1]), N0=int(o[2]), N1=int(o[3]), tz0=int(o[4]), tz1=int(o[5]))


def deinterleave_bgr(img_u8):
    flat = np.ascontiguousarray(img_u8, np.uint8).reshape(-1)
    n = flat.size // 3
    out = np.empty((3, n), np.float32)
    lib().ora_deinterleave_bgr_u8_f32(flat, out[0], out[1], out[2], n)
    return out


def interleave_bgr(planes_f32):
    p = np.ascontiguousarray(planes_f32, np.float32).reshape(3, -1)
    n = p.shape[1]
    out = np.empty(n * 3, np.uint8)
    lib().ora_interleave_bgr_f32_u8(p[0], p[1], p[2], out, n)
    return out


def reflect_101(img, pt, pb, pl, pr):
    a = np.ascontiguousarray(img)
    rows, cols = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    pt, pb = min(pt, rows - 1), min(pb, rows - 1)
    pl, pr = min(pl, cols - 1), min(pr, cols - 1)
    out = np.empty((rows + pt + pb, cols + pl + pr) + (() if a.ndim == 2 else (ch,)), a.dtype)
    pads = np.zeros(4, np.int32)
    lib().ora_reflect_101(a.ctypes.data, out.ctypes.data, a.dtype.itemsize, ch, pt, pb, pl, pr,
                          np.array([rows, cols], np.int32), pads)
    return out


def flip_block(plane, w, h):
    a = np.ascontiguousarray(plane, np.float32).reshape(-1)
    out = np.empty_like(a)
    lib().ora_flip_block_f32(a, out, w, h)
    return out


def pad_tile(x, pad, n):
    x = np.ascontiguousarray(x, np.float32)
    t = np.empty(n, np.float32)
    lib().ora_pad_tile(x, x.size, pad, n, t)
    return t


def fft64(x, backward=False):
    x = np.ascontiguousarray(x, np.complex128)
    out = np.empty_like(x)
    lib().ora_fft64(x.size, x.view(np.float64), out.view(np.float64), int(backward))
    return out


def kernel_multipliers(sigma, ksize, n):
    m = np.empty(n // 2 + 1, np.float32)
    lib().ora_kernel_multipliers(float(sigma), ksize, n, m)
    return m


def pffft_plane_f64(plane, sigma, quirk=True, want_inter=False):
    """per-channel body of pffft_() (Source.cpp:510-564), float64 arithmetic."""
    p = np.array(plane, np.float32, order="C")
    rows, cols = p.shape
    inter = np.empty((rows, cols), np.float32) if want_inter else None
    rc = lib().ora_pffft_plane_f64(p, rows, cols, float(sigma), int(quirk),
                                   inter.ctypes.data if want_inter else None)
    if rc:
        raise ValueError("pad > min(rows, cols) - 1")
    return (p, inter) if want_inter else p


def pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=False):
    """pffft_(image, sigma) (Source.cpp:429-570), float64 arithmetic."""
    a = np.ascontiguousarray(img, np.uint8)
    rows, cols, ch = a.shape
    assert ch == 3
    out = np.empty_like(a)
    planes = np.empty((3, rows, cols), np.float32) if want_planes else None
    rc = lib().ora_pffft_blur_u8c3_f64(a.reshape(-1), out.reshape(-1), rows, cols, float(sigma), int(quirk),
                                       planes.ctypes.data if want_planes else None)
    if rc:
        raise ValueError("pad > min(rows, cols) - 1")
    return (out, planes) if want_planes else out


def pffft_blur_u8c3_f32(img, sigma):
    """float32 port with the reference's stage structure (the timed CPU baseline)."""
    a = np.ascontiguousarray(img, np.uint8)
    rows, cols, ch = a.shape
    assert ch == 3
    out = np.empty_like(a)
    rc = lib().ora_pffft_blur_u8c3_f32(a.reshape(-1), out.reshape(-1), rows, cols, float(sigma))
    if rc:
        raise ValueError("pad > min(rows, cols) - 1")
    return out


class RealFFT:
    """pffft-shaped real transform of the float32 port (ordered layout)."""

    def __init__(self, n):
        self.n = n
        self.s = lib().ora_fft_new_setup(n)
        if not self.s:
            raise ValueError("unsupported length %d" % n)

    def __del__(self):
        if getattr(self, "s", None):
            lib().ora_fft_destroy_setup(self.s)
            self.s = None

    def transform_ordered(self, x, backward=False):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(self.n, np.float32)
        work = np.empty(self.n, np.float32)
        lib().ora_fft_transform_ordered(self.s, x, out, work, int(backward))
        return out


def sorted_optimized_convolution(tile_dft, kernel_dft, scaler):
    t = np.array(tile_dft, np.float32)
    lib().ora_sorted_optimized_convolution(t, np.ascontiguousarray(kernel_dft, np.float32), t.size, float(scaler))
    return t


def fastboxblur_u8(img, ksize, passes):
    a = np.array(img, np.uint8, order="C")
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    rc = lib().ora_fastboxblur_u8(a.reshape(-1), w, h, ch, int(ksize), int(passes))
    if rc:
        raise ValueError("bad fastboxblur arguments")
    return a


def box_kernel_1d(klen, fft_length):
    """box_kernel (1D), Source.cpp:129-140: tent kernel centred at index 0 of an N-periodic array"""
    k = np.zeros(fft_length, np.float32)
    lib().ora_box_kernel_1d(k, int(klen), int(fft_length))
    return k


def boxfft_sizing(rows, cols, nsmooth):
    o = np.zeros(4, np.int32)
    lib().ora_boxfft_sizing(rows, cols, float(nsmooth), o)
    return dict(kLen=int(o[0]), pad=int(o[1]), N0=int(o[2]), N1=int(o[3]))


def pffft_blur_u8c3_f64_kernel(img, pad, kern_row, kern_col, quirk=True, want_planes=False):
    """pffft_() with caller-supplied N-periodic kernels (centre at index 0) and pad, float64 arithmetic"""
    a = np.ascontiguousarray(img, np.uint8)
    rows, cols, _ = a.shape
    out = np.empty_like(a)
    planes = np.empty((3, rows, cols), np.float32) if want_planes else None
    rc = lib().ora_pffft_blur_u8c3_f64_kernel(a.reshape(-1), out.reshape(-1), rows, cols, int(pad),
                                              np.ascontiguousarray(kern_row, np.float32), np.ascontiguousarray(kern_col, np.float32),
                                              int(quirk), planes.ctypes.data if want_planes else None)
    if rc:
        raise ValueError("pad > min(rows, cols) - 1")
    return (out, planes) if want_planes else out


def pffft_boxblur_u8c3_f64(img, nsmooth, quirk=True, want_planes=False):
    """the `#define boxblur` branch of pffft_() (Source.cpp:437-442,468-472): FFT-domain tent kernel"""
    rows, cols, _ = img.shape
    s = boxfft_sizing(rows, cols, nsmooth)
    return pffft_blur_u8c3_f64_kernel(img, s["pad"], box_kernel_1d(s["kLen"], s["N1"]), box_kernel_1d(s["kLen"], s["N0"]), quirk, want_planes)


def num_threads():
    return lib().ora_num_threads()


# --------------------------------------------------------------------------
# the reference's own code (oracle/_ref), when the prebuilt library is present
# --------------------------------------------------------------------------
_ref = None


def ref_available():
    build()
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(_REF)
        R.ref_gaussian_window.argtypes = [C.c_double, C.c_int]
        R.ref_get_gaussian.argtypes = [_f32p, C.c_double, C.c_int, C.c_int]
        R.ref_is_valid_size.argtypes = [C.c_int]
        R.ref_nearest_transform_size.argtypes = [C.c_int]
        R.ref_deinterleave_bgr_u8_f32.argtypes = [_u8p, _f32p, _f32p, _f32p, C.c_uint32]
        R.ref_interleave_bgr_f32_u8.argtypes = [_f32p, _f32p, _f32p, _u8p, C.c_uint32]
        for nm in ("ref_reflect_101_u8c3", "ref_reflect_101_u8c1", "ref_reflect_101_f32c1"):
            getattr(R, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
        R.ref_hybrid_loop_count.argtypes = [C.c_int, _i32p]
        if hasattr(R, "ref_sorted_optimized_convolution"):
            R.ref_sorted_optimized_convolution.argtypes = [_f32p, _f32p, C.c_int, C.c_float]
        _ref = R
    return _ref


def ref_get_gaussian(sigma, width=0, fft_length=0):
    w = width or ref().ref_gaussian_window(float(sigma), 0)
    k = np.zeros(max(w, fft_length), np.float32)
    ref().ref_get_gaussian(k, float(sigma), int(width), int(fft_length))
    return k


def ref_sorted_optimized_convolution(tile_dft, kernel_dft, scaler):
    """the reference's own pffft_sorted_optimized_convolution (Source.cpp:414-427), compiled from where it lies"""
    t = np.array(tile_dft, np.float32)
    ref().ref_sorted_optimized_convolution(t, np.ascontiguousarray(kernel_dft, np.float32), t.size, float(scaler))
    return t


def ref_reflect_101(img, pt, pb, pl, pr):
    a = np.ascontiguousarray(img)
    rows, cols = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    fn = {(np.dtype(np.uint8), 3): "ref_reflect_101_u8c3", (np.dtype(np.uint8), 1): "ref_reflect_101_u8c1",
          (np.dtype(np.float32), 1): "ref_reflect_101_f32c1"}[(a.dtype, ch)]
    cpt, cpb = min(pt, rows - 1), min(pb, rows - 1)
    cpl, cpr = min(pl, cols - 1), min(pr, cols - 1)
    out = np.zeros((rows + cpt + cpb, cols + cpl + cpr) + (() if a.ndim == 2 else (ch,)), a.dtype)
    getattr(ref(), fn)(a.ctypes.data, out.ctypes.data, pt, pb, pl, pr, np.array([rows, cols], np.int32))
    return out
