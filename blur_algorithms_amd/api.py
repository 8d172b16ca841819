"""Host-side mirror of the reference's call surface for the FFT-blur hot path.

Names and argument meaning follow michelerenzullo/Blur_algorithms (Source.cpp / Utils.hpp);
the work happens in libblur_amd.so (hand-written HIP for gfx950) through the C ABI of
include/blur_amd.h.  torch is used only to hold device memory and the stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BlurError, BlurOpts


def _L():
    return _lib.load()


# ---- sizing (host, bit-identical to the reference) -------------------------------------
def gaussian_window(sigma, max_width=0):
    """gaussian_window(sigma, max_width) -- Source.cpp:60-73"""
    return _L().blur_gaussian_window(float(sigma), int(max_width))


def getGaussian(sigma, width=0, FFT_length=0):
    """getGaussian(kernel, sigma, width, FFT_length) -- Source.cpp:75-102; returns the kernel"""
    w = width or gaussian_window(sigma)
    k = np.zeros(max(w, FFT_length), np.float32)
    rc = _L().blur_get_gaussian(k.ctypes.data, float(sigma), int(width), int(FFT_length))
    if rc:
        raise BlurError(rc, "getGaussian: bad arguments")
    return k


def isValidSize(N):
    """isValidSize -- Utils.hpp:141-148"""
    return _L().blur_is_valid_size(int(N))


def nearestTransformSize(N):
    """nearestTransformSize -- Utils.hpp:150-157"""
    return _L().blur_nearest_transform_size(int(N))


def pffft_sizing(rows, cols, sigma):
    """the sizing block of pffft_() -- Source.cpp:434-457"""
    out = (C.c_int * 6)()
    rc = _L().blur_pffft_sizing(int(rows), int(cols), float(sigma), out)
    if rc:
        raise BlurError(rc, "pffft_sizing: bad arguments")
    return dict(kSize=out[0], pad=out[1], N0=out[2], N1=out[3], tz0=out[4], tz1=out[5])


def kernel_multipliers(sigma, ksize, n):
    m = np.empty(n // 2 + 1, np.float32)
    rc = _L().blur_kernel_multipliers(float(sigma), int(ksize), int(n), m.ctypes.data)
    if rc:
        raise BlurError(rc, "kernel_multipliers: bad arguments")
    return m


def box_kernel(kLen, FFT_length):
    """box_kernel(kernel, kLen, FFT_length), 1D form -- Source.cpp:129-140; returns the kernel"""
    k = np.zeros(int(FFT_length), np.float32)
    rc = _L().blur_box_kernel(k.ctypes.data, int(kLen), int(FFT_length))
    if rc:
        raise BlurError(rc, "box_kernel: bad arguments")
    return k


def pocketfft2d_sizing(rows, cols, sigma):
    """the sizing block of pocketfft_2D -- Source.cpp:149-176"""
    out = (C.c_int * 8)()
    rc = _L().blur_pocketfft2d_sizing(int(rows), int(cols), float(sigma), out)
    if rc:
        raise BlurError(rc, "pocketfft2d_sizing: bad arguments")
    return dict(kSize=out[0], pad=out[1], sizes=(out[2], out[3]), border=(out[4], out[5], out[6], out[7]))


def boxfft_sizing(rows, cols, nsmooth):
    """sizing of the `#define boxblur` mode of pffft_() -- Source.cpp:437-457"""
    out = (C.c_int * 4)()
    rc = _L().blur_boxfft_sizing(int(rows), int(cols), float(nsmooth), out)
    if rc:
        raise BlurError(rc, "boxfft_sizing: bad arguments")
    return dict(kLen=out[0], pad=out[1], N0=out[2], N1=out[3])


def fft_plan_radices(n):
    r = (C.c_int * 16)()
    k = _L().blur_fft_plan_radices(int(n), r)
    return [r[i] for i in range(k)]


# ---- the GPU context ------------------------------------------------------------------
# enum blur_engine (include/blur_amd.h)
ENGINES = {"auto": 0, "rows-first": 1, "wave-resident": 2, "matrix": 3, "fft": 5, "fused": 6}


class BlurContext:
    """Owns the plan / kernel-spectrum caches and the float32 workspace on one GPU
    (role of the PFFFT_Setup pair the reference rebuilds per call, Source.cpp:477-478)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._lib = _L()
        rc = self._lib.blur_ctx_create(C.byref(self._h), int(device))
        if rc:
            raise BlurError(rc, self._lib.blur_last_error(None).decode())
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.blur_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise BlurError(rc, self._lib.blur_last_error(self._h).decode())

    def _opts(self, nyquist_quirk=True, col_group=0, force_generic=False, frames_per_launch=0, row_major_planes=False,
              wave_resident=None, engine=None, tile_points=0):
        # one blur_opts per distinct argument tuple, built once (a call on a small image is tens of microseconds of GPU time: the
        # wrapper must not cost more than the kernels)
        key = (bool(nyquist_quirk), int(col_group), bool(force_generic), int(frames_per_launch), bool(row_major_planes), wave_resident, engine, int(tile_points))
        cache = self.__dict__.setdefault("_opts_cache", {})
        o = cache.get(key)
        if o is not None:
            return o
        o = cache[key] = self._build_opts(*key)
        return o

    def _build_opts(self, nyquist_quirk, col_group, force_generic, frames_per_launch, row_major_planes, wave_resident, engine, tile_points=0):
        o = BlurOpts()
        self._lib.blur_opts_default(C.byref(o))
        o.nyquist_quirk = 1 if nyquist_quirk else 0
        o.col_group = int(col_group)
        o.force_generic = 1 if force_generic else 0   # tests: run the run-time-planned kernels even where a specialised one exists
        o.frames_per_launch = int(frames_per_launch)
        o.row_major_planes = 1 if row_major_planes else 0
        # wave-resident kernels (transform length 256 * R0, columns first): None = where they pay (the image fills most
        # of the transform), False = never, True = wherever the image fits one
        o.engine = 0 if wave_resident is None else (2 if wave_resident else 1)
        # engine (enum blur_engine): None = the library's choice (the fused matrix-core kernel where it applies, else the two-kernel
        # matrix-core engine, else the FFT kernels); "fused" / "matrix" = Toeplitz products on the f16 matrix cores in one kernel
        # (fx_kernels.hpp) / two (mx_kernels.hpp); "fft" = the FFT kernels with their own measured choice of family;
        # "wave-resident" / "rows-first" = one FFT family
        if engine is not None:
            o.engine = ENGINES[engine]
        o.tile_points = int(tile_points)      # tests: force the tiled wave-resident path with transforms of at most this many points
        return o

    def use_torch_stream(self):
        """launch on torch's current stream of this device (the raw handle: torch.cuda.current_stream() builds a Stream object per
        call, several microseconds; the context is only told when the handle changed)"""
        import torch
        try:
            h = torch._C._cuda_getCurrentRawStream(self.device)
        except AttributeError:  # pragma: no cover - older / newer torch without the private accessor
            h = torch.cuda.current_stream(self.device).cuda_stream
        if h != self.__dict__.get("_stream_handle", -1):
            self._check(self._lib.blur_ctx_set_stream(self._h, C.c_void_p(h)))
            self._stream_handle = h

    def set_stream(self, handle):
        self._check(self._lib.blur_ctx_set_stream(self._h, C.c_void_p(handle)))
        self._stream_handle = handle

    def synchronize(self):
        self._check(self._lib.blur_ctx_synchronize(self._h))

    def last_family(self):
        """kernels the last u8c3 blur ran on: 0 run-time plans, 1 rows-first, 2 wave-resident, 3 whole-image 2D, 4 matrix-core
        (two kernels), 6 fused matrix-core (debug query, not declared in the public header)"""
        fn = self._lib.blur_debug_last_family
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_int
        return int(fn(self._h))

    def last_engine(self):
        """(family code, note): the kernels the last u8c3 blur ran on and, under the library's own choice, why a faster engine was
        passed over (blur_last_engine)"""
        buf = C.create_string_buffer(512)
        fam = int(self._lib.blur_last_engine(self._h, buf, 512))
        return fam, buf.value.decode()

    def copy_bandwidth(self, mib=1024, reps=5):
        """GB/s (read + written) of a 16-byte-per-lane device copy of `mib` MiB: the box's streaming rate for the kernels' access shape"""
        g = C.c_double(0)
        self._check(self._lib.blur_copy_bandwidth(self._h, int(mib) << 20, int(reps), C.byref(g)))
        return g.value

    def timing_enable(self, on=True):
        """per-kernel HIP events on the launch stream: True / 1 = every timed launch, 2 = slot 0 only (the dominant kernel), False = off"""
        self._check(self._lib.blur_ctx_timing_enable(self._h, 2 if on == 2 and on is not True else (1 if on else 0)))

    def timing(self, reset=True):
        ms = (C.c_double * 2)()
        n = (C.c_int * 2)()
        fr = (C.c_int * 2)()
        self._check(self._lib.blur_ctx_timing(self._h, ms, n, fr, 1 if reset else 0))
        return dict(row_ms=ms[0], col_ms=ms[1], row_launches=n[0], col_launches=n[1], row_frames=fr[0], col_frames=fr[1])

    # -- pffft_(image, sigma): Source.cpp:429-570 -----------------------------------------
    def pffft_(self, image, sigma, out=None, nyquist_quirk=True, col_group=0, force_generic=False, frames_per_launch=0,
               row_major_planes=False, wave_resident=None, engine=None, tile_points=0):
        """Gaussian blur of a BGR/RGB uint8 image [rows, cols, 3] or a batch [n, rows, cols, 3].

        torch CUDA tensor: asynchronous on torch's current stream, returns `out`
        (default: in place, like the reference).  numpy array: host round trip, returns a new array.
        """
        o = self._opts(nyquist_quirk, col_group, force_generic, frames_per_launch, row_major_planes, wave_resident, engine, tile_points)
        if isinstance(image, np.ndarray):
            if (image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3 and not image.flags["C_CONTIGUOUS"]
                    and image.strides[2] == 1 and image.strides[1] == 3 and image.strides[0] >= 3 * image.shape[1]):
                # a view with padded rows (a cv::Mat ROI): pitched entry point, no host-side repacking
                res = np.empty(image.shape, np.uint8)
                self._check(self._lib.blur_gaussian_u8c3_host_pitched(self._h, image.ctypes.data, image.strides[0], res.ctypes.data,
                                                                      res.strides[0], image.shape[0], image.shape[1], float(sigma), C.byref(o)))
                return res
            a = np.ascontiguousarray(image, np.uint8)
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("expected a uint8 image of shape [rows, cols, 3]")
            res = np.empty_like(a)
            self._check(self._lib.blur_gaussian_u8c3_host(self._h, a.ctypes.data, res.ctypes.data, a.shape[0], a.shape[1],
                                                          float(sigma), C.byref(o)))
            return res
        import torch
        t = image
        if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.shape[-1] != 3 or t.dim() not in (3, 4):
            raise ValueError("expected a contiguous CUDA uint8 tensor [rows, cols, 3] or [n, rows, cols, 3]")
        dst = t if out is None else out
        if dst.shape != t.shape or dst.dtype != t.dtype or not dst.is_cuda or not dst.is_contiguous():
            raise ValueError("out must match the input")
        self.use_torch_stream()
        n = 1 if t.dim() == 3 else t.shape[0]
        rows, cols = t.shape[-3], t.shape[-2]
        self._check(self._lib.blur_gaussian_u8c3_batch_dev(self._h, t.data_ptr(), dst.data_ptr(), n, rows, cols, float(sigma), C.byref(o)))
        return dst

    def convolve_lines(self, lines, multipliers, out=None):
        """lines: CUDA complex64 tensor [nlines, n]; multipliers: n real factors (numpy float32, natural frequency order).
        Returns IDFT(multipliers * DFT(line)) per line, unnormalised (blur_convolve_lines_c32_dev): the batched form of
        pffft_transform_ordered / pffft_sorted_optimized_convolution / pffft_transform_ordered (Source.cpp:531-533)."""
        import torch
        t = lines
        if t.dtype != torch.complex64 or not t.is_cuda or not t.is_contiguous() or t.dim() != 2:
            raise ValueError("expected a contiguous CUDA complex64 tensor [nlines, n]")
        dst = torch.empty_like(t) if out is None else out
        m = np.ascontiguousarray(multipliers, np.float32)
        if m.shape != (t.shape[1],):
            raise ValueError("one multiplier per frequency")
        self.use_torch_stream()
        self._check(self._lib.blur_convolve_lines_c32_dev(self._h, t.data_ptr(), dst.data_ptr(), t.shape[0], t.shape[1], m.ctypes.data))
        return dst

    def pinned_empty(self, shape, dtype=np.uint8):
        """numpy array in page-locked host memory (blur_host_alloc); freed when the array and its views are gone"""
        import weakref
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._lib.blur_host_alloc(self._h, C.byref(p), n))
        raw = (C.c_uint8 * max(n, 1)).from_address(p.value)
        lib, h, addr = self._lib, self._h, p.value
        weakref.finalize(raw, lambda: lib.blur_host_free(h, addr))
        return np.frombuffer(raw, np.uint8, n).view(dtype).reshape(shape)

    def pffft_host_batch(self, frames, sigma, out=None, nyquist_quirk=True):
        """frames: numpy uint8 [n, rows, cols, 3] in host memory (pinned_empty() for full PCIe overlap);
        copies in, kernels and copies out are pipelined over three device slots.  Returns `out` (default: a new array)."""
        a = frames
        if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.ndim == 4 and a.shape[3] == 3 and a.flags["C_CONTIGUOUS"]):
            raise ValueError("expected a contiguous uint8 array [n, rows, cols, 3]")
        res = np.empty_like(a) if out is None else out
        if res.shape != a.shape or res.dtype != np.uint8 or not res.flags["C_CONTIGUOUS"]:
            raise ValueError("out must match the input")
        o = self._opts(nyquist_quirk)
        self._check(self._lib.blur_gaussian_u8c3_host_batch(self._h, a.ctypes.data, res.ctypes.data, a.shape[0], a.shape[1], a.shape[2],
                                                            float(sigma), C.byref(o)))
        return res

    def pocketfft_1D(self, image, sigma, out=None, **kw):
        """pocketfft_1D(image, sigma) (Source.cpp:280-392): the same 1D tiles as pffft_() with N/2+1 bins and the
        true Nyquist multiplier (:362,:378) -- this engine's `nyquist_quirk = 0` mode."""
        return self.pffft_(image, sigma, out=out, nyquist_quirk=False, **kw)

    def pocketfft_2D(self, image, sigma, out=None, **kw):
        """pocketfft_2D(image, sigma) (Source.cpp:143-277): Reflect_101 of the whole image, 2D r2c, separable
        multiply, c2r, crop.  Inside the crop that is the linear convolution of the reflect-101 extended image with
        the same taps (every border is >= the kernel half width, so neither the wrap-around nor the extra
        transform-size padding reaches a kept pixel): the 1D-tiled engine computes it without materialising the
        padded image.  tests/ compare with the scipy.fft (pocketfft) restatement of the 2D path.

        whole_image=True runs the reference's own structure instead (blur_pocketfft2d_u8c3_dev: the padded image as
        ONE 2D transform with pocketfft_2D's sizes and borders); want_planes=True then also returns the float planes
        [3, rows, cols] before the "+0.5f, truncate"."""
        if kw.pop("whole_image", False):
            return self._pocketfft2d(image, sigma, out, 0, kw.pop("want_planes", False))
        return self.pffft_(image, sigma, out=out, nyquist_quirk=False, **kw)

    def DFT_image(self, image, sigma, out=None, want_planes=False):
        """pocketfft_2D compiled with `#define DFT_image` (Source.cpp:235-252): every plane becomes the fft-shifted
        log spectrum 20 log10(|Re F| + 1e-5) of the reflect-101 padded plane (sigma only sets the padding), read with
        the reference's index arithmetic, interleaved ("+0.5f, truncate") and cropped like the blur."""
        return self._pocketfft2d(image, sigma, out, 1, want_planes)

    def _pocketfft2d(self, image, sigma, out, dft_image, want_planes):
        if isinstance(image, np.ndarray):                       # host round trip, returns a new array
            a = np.ascontiguousarray(image, np.uint8)
            if a.ndim != 3 or a.shape[2] != 3 or want_planes:
                raise ValueError("expected a uint8 image [rows, cols, 3] (float planes: device tensors only)")
            res = np.empty_like(a)
            self._check(self._lib.blur_pocketfft2d_u8c3_host(self._h, a.ctypes.data, res.ctypes.data, a.shape[0], a.shape[1], float(sigma), int(dft_image)))
            return res
        import torch
        if image.dtype != torch.uint8 or not image.is_cuda or not image.is_contiguous() or image.dim() != 3 or image.shape[-1] != 3:
            raise ValueError("expected a contiguous CUDA uint8 tensor [rows, cols, 3]")
        dst = image if out is None else out
        planes = torch.empty((3, image.shape[0], image.shape[1]), dtype=torch.float32, device=image.device) if want_planes else None
        self.use_torch_stream()
        self._check(self._lib.blur_pocketfft2d_u8c3_dev(self._h, image.data_ptr(), dst.data_ptr(), image.shape[0], image.shape[1], float(sigma),
                                                        int(dft_image), planes.data_ptr() if want_planes else None))
        return (dst, planes) if want_planes else dst

    def Reflect_101(self, image, top, bottom, left, right):
        """Reflect_101<uint8_t, C> (Utils.hpp:212-243): uint8 CUDA tensor [rows, cols, C] -> padded tensor; borders are
        clamped to dim - 1 like the reference"""
        import torch
        rows, cols, ch = image.shape
        size = (C.c_int * 2)()
        self._check(self._lib.blur_reflect101_u8_dev(self._h, None, None, rows, cols, ch, int(top), int(bottom), int(left), int(right), size))
        out = torch.empty((size[0], size[1], ch), dtype=torch.uint8, device=image.device)
        self.use_torch_stream()
        self._check(self._lib.blur_reflect101_u8_dev(self._h, image.data_ptr(), out.data_ptr(), rows, cols, ch, int(top), int(bottom), int(left), int(right), size))
        return out

    def pffft_boxblur(self, image, nsmooth, out=None, nyquist_quirk=True):
        """pffft_() compiled with `#define boxblur`: FFT-domain tent kernel (Source.cpp:437-442,468-472);
        uint8 CUDA tensor [rows, cols, 3]"""
        o = self._opts(nyquist_quirk)
        dst = image if out is None else out
        self.use_torch_stream()
        self._check(self._lib.blur_boxfft_u8c3_dev(self._h, image.data_ptr(), dst.data_ptr(), image.shape[0], image.shape[1],
                                                   float(nsmooth), C.byref(o)))
        return dst

    def separable(self, image, taps, pad=None, out=None, nyquist_quirk=True, engine=None):
        """any symmetric separable kernel (odd tap count) through the same engines; pad defaults to len(taps)//2"""
        o = self._opts(nyquist_quirk, engine=engine)
        t = np.ascontiguousarray(taps, np.float32)
        dst = image if out is None else out
        self.use_torch_stream()
        self._check(self._lib.blur_separable_u8c3_dev(self._h, image.data_ptr(), dst.data_ptr(), image.shape[0], image.shape[1],
                                                      t.ctypes.data, t.size, t.size // 2 if pad is None else int(pad), C.byref(o)))
        return dst

    def pffft_plane(self, plane, sigma, out=None, nyquist_quirk=True, col_group=0):
        """the per-channel body of pffft_() on one float32 plane (Source.cpp:510-564)"""
        o = self._opts(nyquist_quirk, col_group)
        if isinstance(plane, np.ndarray):
            a = np.ascontiguousarray(plane, np.float32)
            res = np.empty_like(a)
            self._check(self._lib.blur_gaussian_f32c1_host(self._h, a.ctypes.data, res.ctypes.data, a.shape[0], a.shape[1],
                                                           float(sigma), C.byref(o)))
            return res
        import torch
        t = plane
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.dim() != 2:
            raise ValueError("expected a contiguous CUDA float32 tensor [rows, cols]")
        dst = torch.empty_like(t) if out is None else out
        self.use_torch_stream()
        self._check(self._lib.blur_gaussian_f32c1_dev(self._h, t.data_ptr(), dst.data_ptr(), t.shape[0], t.shape[1], float(sigma), C.byref(o)))
        return dst

    def rowpass(self, image, sigma, nyquist_quirk=True, force_generic=False, engine=None):
        """row pass only: uint8 [rows, cols, 3] CUDA tensor -> float32 [3, rows, cols] (Source.cpp:520-537); engine="fused": the planes the
        fused matrix-core kernel hands from its row pass to its column pass (a test build of the kernel writes them out)"""
        import torch
        o = self._opts(nyquist_quirk, 0, force_generic, engine=engine)
        rows, cols = image.shape[0], image.shape[1]
        planes = torch.empty((3, rows, cols), dtype=torch.float32, device=image.device)
        self.use_torch_stream()
        self._check(self._lib.blur_rowpass_u8c3_dev(self._h, image.data_ptr(), planes.data_ptr(), rows, cols, float(sigma), C.byref(o)))
        return planes

    # -- the pieces either side ----------------------------------------------------------------
    def flip_block(self, plane, w, h):
        """flip_block<float,1>(in, out, w, h) -- call sites Source.cpp:540,562"""
        import torch
        out = torch.empty_like(plane)
        self.use_torch_stream()
        self._check(self._lib.blur_flip_block_f32_dev(self._h, plane.data_ptr(), out.data_ptr(), int(w), int(h)))
        return out

    def deinterleave_BGR(self, image):
        """deinterleave_BGR<uint8_t,float> -- Utils.hpp:159-184; returns float32 [3, total]"""
        import torch
        total = image.numel() // 3
        planes = torch.empty((3, total), dtype=torch.float32, device=image.device)
        self.use_torch_stream()
        self._check(self._lib.blur_deinterleave_bgr_u8_f32_dev(self._h, image.data_ptr(), planes.data_ptr(), total))
        return planes

    def interleave_BGR(self, planes):
        """interleave_BGR<uint8_t,float> -- Utils.hpp:186-210; planes float32 [3, total]"""
        import torch
        total = planes.shape[1]
        out = torch.empty(total * 3, dtype=torch.uint8, device=planes.device)
        self.use_torch_stream()
        self._check(self._lib.blur_interleave_bgr_f32_u8_dev(self._h, planes.data_ptr(), out.data_ptr(), total))
        return out

    def fastboxblur(self, image, ksize, passes):
        """fastboxblur(in, w, h, channels, ksize, passes), in place -- call site Source.cpp:587"""
        if isinstance(image, np.ndarray):
            a = np.array(image, np.uint8, order="C")
            h, w = a.shape[:2]
            ch = 1 if a.ndim == 2 else a.shape[2]
            self._check(self._lib.blur_fastboxblur_u8_host(self._h, a.ctypes.data, w, h, ch, int(ksize), int(passes)))
            return a
        h, w = image.shape[0], image.shape[1]
        ch = 1 if image.dim() == 2 else image.shape[2]
        self.use_torch_stream()
        self._check(self._lib.blur_fastboxblur_u8_dev(self._h, image.data_ptr(), w, h, ch, int(ksize), int(passes)))
        return image


class BlurMulti:
    """Several GPUs (or several logical shards on one GPU) behind one handle: blur_multi_* of include/blur_amd.h.
    Frames of a batch are sharded by frame, one context and stream per shard, driven from this one host thread."""

    def __init__(self, devices):
        self._lib = _L()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        rc = self._lib.blur_multi_create(C.byref(self._h), devs, len(devices))
        if rc:
            raise BlurError(rc, "blur_multi_create failed")
        self.devices = list(devices)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.blur_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise BlurError(rc, self._lib.blur_multi_last_error(self._h).decode())

    def pffft_(self, frames, sigma, out=None, nyquist_quirk=True, wave_resident=None):
        """frames: uint8 [n, rows, cols, 3]; a torch CUDA tensor on devices[0] or a numpy array in host memory.  Synchronous."""
        o = BlurOpts()
        self._lib.blur_opts_default(C.byref(o))
        o.nyquist_quirk = 1 if nyquist_quirk else 0
        o.engine = 0 if wave_resident is None else (2 if wave_resident else 1)
        if isinstance(frames, np.ndarray):
            a = np.ascontiguousarray(frames, np.uint8)
            if a.ndim != 4 or a.shape[3] != 3:
                raise ValueError("expected uint8 frames [n, rows, cols, 3]")
            res = np.empty_like(a) if out is None else out
            self._check(self._lib.blur_gaussian_u8c3_batch_multi_host(self._h, a.ctypes.data, res.ctypes.data, a.shape[0], a.shape[1], a.shape[2],
                                                                      float(sigma), C.byref(o)))
            return res
        import torch
        t = frames
        if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.dim() != 4 or t.shape[-1] != 3 or t.device.index != self.devices[0]:
            raise ValueError("expected a contiguous CUDA uint8 tensor [n, rows, cols, 3] on devices[0]")
        dst = t if out is None else out
        torch.cuda.synchronize(t.device)
        self._check(self._lib.blur_gaussian_u8c3_batch_multi_dev(self._h, t.data_ptr(), dst.data_ptr(), t.shape[0], t.shape[1], t.shape[2],
                                                                 float(sigma), C.byref(o)))
        return dst

