"""oracle/pocketfft_paths.py -- the reference's two pocketfft blur paths, restated with scipy.fft.

TEST INFRASTRUCTURE ONLY (see oracle/blur_oracle.c): imported by tests/ and nothing else.

scipy.fft is built on pocketfft (the C++ header-only library `pocketfft_hdronly.h` that the
reference includes as a submodule, Source.cpp:17; scipy vendors the same library and keeps float32
inputs in float32), so these functions run the reference's pocketfft paths on the third-party
transform the reference itself calls -- an anchor for the engine's `nyquist_quirk = 0` mode that
does not depend on this repo's own FFT restatement.

    pocketfft_1d_u8c3   Source.cpp:280-392   1D tiles, same structure as pffft_() but N/2+1 bins
                                             with the true Nyquist multiplier (:362,:378)
    pocketfft_2d_u8c3   Source.cpp:143-277   whole padded image, r2c over both axes, crop (:268-276)
    dft_image_u8c3      Source.cpp:235-252   the same function compiled with `#define DFT_image`: log spectrum image

Both equal, in exact arithmetic, a linear convolution of the reflect-101 extended image inside the
cropped region, which is also what pffft_() computes apart from its Nyquist-slot quirk
(Source.cpp:420-425): the engine serves all three with one data path.
"""
import numpy as np
import scipy.fft as sfft

from . import oracle as O


def _round_u8(planes):
    """interleave_BGR's float -> u8: add 0.5f and truncate (Utils.hpp:189,204-206)"""
    v = np.asarray(planes, np.float32) + np.float32(0.5)
    return np.moveaxis(np.clip(np.trunc(v), 0, 255).astype(np.uint8), 0, -1).copy()


def _sizes_1d(rows, cols, sigma):
    """Source.cpp:283-306"""
    ksize = O.gaussian_window(sigma, max(rows, cols))
    pad = (ksize - 1) // 2
    sizes = [rows + 2 * pad, cols + 2 * pad]
    tz = [0, 0]
    for i in range(2):
        if not O.is_valid_size(sizes[i]):
            n = O.nearest_transform_size(sizes[i])
            tz[i] = n - sizes[i]
            sizes[i] = n
    return ksize, pad, sizes, tz


def _tiles(plane, pad, n):
    """rows of `plane` -> [rows, n] tiles: reversed left border, data, reversed right border, zeros
    (Source.cpp:357-359, :373-375)"""
    rows, length = plane.shape
    t = np.zeros((rows, n), plane.dtype)
    if pad:
        t[:, :pad] = plane[:, pad:0:-1]
        t[:, pad + length:pad + length + pad] = plane[:, length - 2:length - 2 - pad:-1]
    t[:, pad:pad + length] = plane
    return t


def _conv_lines(plane, pad, n, kerf_real):
    """one hybrid_loop body for every line at once: r2c, * real(kerf), c2r with fct 1/n, crop
    (Source.cpp:361-365)"""
    t = _tiles(plane, pad, n)
    w = sfft.rfft(t, axis=1)
    w *= kerf_real[None, :]
    back = sfft.irfft(w, n=n, axis=1)
    return np.ascontiguousarray(back[:, pad:pad + plane.shape[1]]).astype(plane.dtype, copy=False)


def pocketfft_1d_u8c3(img, sigma, dtype=np.float32, want_planes=False):
    """pocketfft_1D(image, sigma), Source.cpp:280-392 (Gaussian build)"""
    a = np.ascontiguousarray(img, np.uint8)
    rows, cols, ch = a.shape
    assert ch == 3
    ksize, pad, sizes, _ = _sizes_1d(rows, cols, sigma)
    if pad > min(rows, cols) - 1:
        raise ValueError("pad > min(rows, cols) - 1")
    # kernel spectra (Source.cpp:325-347): getGaussian at the FFT length, r2c, real part used
    kerf = {n: np.real(sfft.rfft(O.get_gaussian(sigma, ksize, n).astype(dtype))).astype(dtype) for n in set(sizes)}
    planes = np.moveaxis(a, -1, 0).astype(dtype)                      # deinterleave_BGR (:308-310)
    out = np.empty((3, rows, cols), dtype)
    for c in range(3):
        resf = _conv_lines(planes[c], pad, sizes[1], kerf[sizes[1]])  # rows (:353-368)
        colp = _conv_lines(np.ascontiguousarray(resf.T), pad, sizes[0], kerf[sizes[0]])   # flip_block, columns (:369-384)
        out[c] = colp.T                                               # flip_block back (:386)
    u8 = _round_u8(out)
    return (u8, out.astype(np.float32)) if want_planes else u8


def _sizes_2d(rows, cols, sigma):
    """Source.cpp:149-176 -> ksize, pad, sizes, border (top, bottom, left, right)"""
    ksize = O.gaussian_window(sigma, max(rows, cols))
    pad = (ksize - 1) // 2
    border = [pad, pad, pad, pad]
    sizes = [rows + 2 * pad, cols + 2 * pad]
    for i in range(2):
        if not O.is_valid_size(sizes[i]):
            n = O.nearest_transform_size(sizes[i])
            new_pad = n - sizes[i]
            sizes[i] = n
            border[2 * i] += new_pad // 2
            border[2 * i + 1] = int(np.float32(border[2 * i + 1]) + np.float32(new_pad) / np.float32(2) + np.float32(0.5))
    return ksize, pad, sizes, border


def pocketfft_2d_u8c3(img, sigma, dtype=np.float32, want_planes=False):
    """pocketfft_2D(image, sigma), Source.cpp:143-277 (Gaussian build, without DFT_image)"""
    a = np.ascontiguousarray(img, np.uint8)
    rows, cols, ch = a.shape
    assert ch == 3
    ksize, pad, sizes, border = _sizes_2d(rows, cols, sigma)         # (:149-176)
    if max(border[0], border[1]) > rows - 1 or max(border[2], border[3]) > cols - 1:
        raise ValueError("border > dim - 1: Reflect_101 would clamp it and the reference's buffers no longer match")
    padded = O.reflect_101(a, *border)                                # (:178-180)
    assert padded.shape[:2] == (sizes[0], sizes[1])
    planes = np.moveaxis(padded, -1, 0).astype(dtype)                 # deinterleave_BGR (:185-187)
    # kernel spectra; the column one is mirrored to full length around the Nyquist bin (:204-214)
    kc = np.real(sfft.rfft(O.get_gaussian(sigma, ksize, sizes[0]).astype(dtype))).astype(dtype)
    kcol = np.concatenate([kc, kc[1:sizes[0] - sizes[0] // 2][::-1]])
    assert kcol.size == sizes[0]
    krow = np.real(sfft.rfft(O.get_gaussian(sigma, ksize, sizes[1]).astype(dtype))).astype(dtype)
    out = np.empty_like(planes)
    for c in range(3):
        f = sfft.rfft2(planes[c])                                     # (:233)
        f *= (kcol[:, None] * krow[None, :]).astype(dtype)            # (:255-260)
        out[c] = sfft.irfft2(f, s=(sizes[0], sizes[1]))               # fct 1/ndata (:263)
    u8 = _round_u8(out)                                               # interleave_BGR on the padded image (:268)
    crop = (slice(border[0], sizes[0] - border[1]), slice(border[2], sizes[1] - border[3]))
    res = np.ascontiguousarray(u8[crop])                              # (:271-276)
    return (res, np.ascontiguousarray(out[:, crop[0], crop[1]]).astype(np.float32)) if want_planes else res


def dft_image_u8c3(img, sigma, dtype=np.float32):
    """pocketfft_2D(image, sigma) compiled with `#define DFT_image` (Source.cpp:235-252): instead of the multiply and
    the inverse transform, every padded plane is overwritten with 20 log10(|Re resf| + 1e-5) read through the
    reference's fftshift index arithmetic (:239-244, which mirrors the COLUMN index only into the half spectrum);
    interleave_BGR and the crop follow as in the blur.

    Returns (u8 image [rows, cols, 3], float planes [3, rows, cols], |Re F| planes [3, rows, cols]); the last one lets
    a test tell the pixels whose logarithm is well conditioned (|Re F| far above the transform's rounding noise)
    from those that are rounding noise in ANY float32 FFT, the reference's included.  The u8 conversion is C's
    float -> int -> uint8_t (Utils.hpp:189,204-206): values below zero wrap modulo 256."""
    a = np.ascontiguousarray(img, np.uint8)
    rows, cols, ch = a.shape
    assert ch == 3
    _, _, sizes, border = _sizes_2d(rows, cols, sigma)
    if max(border[0], border[1]) > rows - 1 or max(border[2], border[3]) > cols - 1:
        raise ValueError("border > dim - 1")
    s0, s1 = sizes
    padded = O.reflect_101(a, *border)
    planes = np.moveaxis(padded, -1, 0).astype(dtype)
    row = np.arange(s0)
    col = np.arange(s1)
    row_ = (row + (s0 if s0 % 2 == 0 else s0 + 1) // 2) % s0                      # (:240)
    col_ = (col + (s1 if s1 % 2 == 0 else s1 + 1) // 2) % s1                      # (:241)
    cval = np.where(col_ < s1 // 2 + 1, col_, s1 // 2 - col_ % (s1 // 2))         # (:243)
    logs = np.empty((3, s0, s1), np.float32)
    mags = np.empty((3, s0, s1), np.float64)
    for c in range(3):
        resf = sfft.rfft2(planes[c])                                              # (:233)
        re = np.real(resf)[row_[:, None], cval[None, :]]
        mags[c] = np.abs(re)
        logs[c] = (20 * np.log10(np.abs(re).astype(np.float32) + np.float32(0.00001))).astype(np.float32)   # (:245-248)
    v = logs + np.float32(0.5)
    u8 = np.moveaxis(np.trunc(v).astype(np.int64).astype(np.uint8), 0, -1)        # float -> int -> uint8_t
    crop = (slice(border[0], s0 - border[1]), slice(border[2], s1 - border[3]))
    return (np.ascontiguousarray(u8[crop]), np.ascontiguousarray(logs[:, crop[0], crop[1]]),
            np.ascontiguousarray(mags[:, crop[0], crop[1]]))
