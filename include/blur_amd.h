/*
 * blur_amd.h -- C ABI of the MI355X (gfx950) FFT Gaussian-blur engine.
 *
 * This is the drop-in boundary for ONE hot path of michelerenzullo/Blur_algorithms:
 * the 1D-tiled FFT convolution `pffft_(cv::Mat&, double)` (Source.cpp:429-570) and,
 * for BASELINE config 5, `fastboxblur` (call site Source.cpp:587).  The reference has
 * no FFI of its own (README.md:151-152 "Usage and APIs coming soon"); the entry points
 * below are what a binding for that path would bind: plain pointers and sizes, no C++
 * or torch types.  Each one cites the reference interface it replaces.
 *
 * All device work is hand-written HIP for gfx950 inside libblur_amd.so; there is no
 * CPU fallback: a call that cannot run on the GPU returns an error code.
 *
 * Threading: a blur_ctx is thread-compatible (one call at a time per ctx).  Calls on
 * device pointers are ASYNCHRONOUS on the ctx's stream unless stated otherwise.
 */
#ifndef BLUR_AMD_H
#define BLUR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLUR_AMD_VERSION 100

/* status codes (the reference returns void and checks nothing: Utils.hpp:59-65,
   Source.cpp:477-478,612-621) */
enum {
    BLUR_OK = 0,
    BLUR_ERR_INVALID = 1,      /* bad argument (null pointer, non-positive size, sigma <= 0) */
    BLUR_ERR_UNSUPPORTED = 2,  /* pad > min(rows,cols)-1 (UB in the reference, README.md:33-38) or FFT length too long for LDS */
    BLUR_ERR_HIP = 3,          /* a HIP runtime call failed; see blur_last_error() */
    BLUR_ERR_NOMEM = 4
};

typedef struct blur_ctx blur_ctx;

/* blur_opts.engine.  Every engine computes the same blur under the same parity contract (Parity); the choice is
   about speed and about what each engine can hold.  engine.hip: prepare() holds the policy table of BLUR_ENGINE_AUTO. */
enum blur_engine {
    BLUR_ENGINE_AUTO = 0,            /* fused matrix-core kernel where it exists for the kernel's half width (pad <= 72; pad <= 168
                                        on frames of 1 MP and more; any image width, any pointer alignment); else the two-kernel
                                        matrix-core engine (pad <= 168, non-negative taps) except where the FFT engine has a faster
                                        compile-time family (small frames, the widest kernels on 4K frames); else FFT.  The choice
                                        depends on rows, cols and the kernel only: never on the number of frames, on where the
                                        frames lie in memory or on the environment (blur_last_engine() names it) */
    BLUR_ENGINE_FFT_ROWS_FIRST = 1,  /* FFT kernels, never the wave-resident family */
    BLUR_ENGINE_FFT_WAVE_RESIDENT = 2, /* FFT kernels, wave-resident (transform length 256 R0, columns first) wherever the image fits */
    BLUR_ENGINE_MATRIX = 3,          /* two-kernel matrix-core engine (mx_kernels.hpp); BLUR_ERR_UNSUPPORTED if it cannot hold the kernel */
    BLUR_ENGINE_FFT = 5,             /* FFT kernels with their own measured choice of family */
    BLUR_ENGINE_FUSED = 6            /* fused matrix-core kernels (fx_kernels.hpp, pad <= 72; fw_kernels.hpp, pad <= 168); BLUR_ERR_UNSUPPORTED
                                        where they do not apply */
};

/* Options of the whole-image blur.  Zero-initialise, then blur_opts_default(). */
typedef struct blur_opts {
    /* 1 (default): reproduce pffft_sorted_optimized_convolution exactly
       (Source.cpp:420-425): the Nyquist bin, packed in slot 1 of pffft's ordered
       layout, is scaled with the kernel's DC gain.  0: scale it with the kernel's
       Nyquist gain (what pocketfft_1D does, Source.cpp:362,378). */
    int nyquist_quirk;
    /* columns per workgroup of the column pass (0 = auto: 8 or less as LDS allows) */
    int col_group;
    /* 1: use the run-time-planned FFT kernels even where a compile-time specialised one exists (tests) */
    int force_generic;
    /* > 0: frames per launch pair of the batch entry point (0 = auto: as many as fit a 1 GiB float workspace;
       the fused matrix-core engine has no workspace and takes the whole batch in one launch) */
    int frames_per_launch;
    /* 1: keep the FFT engine's float intermediate in row-major planes even when both passes are specialised
       (default: strips of 8 columns stored contiguously) */
    int row_major_planes;
    /* which kernels run the u8c3 blur: one of enum blur_engine; 0 = the library's choice */
    int engine;
    /* tests: > 0 forces the tiled wave-resident path (bands of rows / tiles of columns through the wave-resident kernels, the quirk
       as rank-one terms) with no transform longer than this many points; 0 = the library decides (lines too long for one transform) */
    int tile_points;
    int reserved[1];   /* must be zero */
} blur_opts;

void blur_opts_default(blur_opts* o);

/* ---- host-side sizing: bit-identical replacements, no GPU needed ------------------ */

/* gaussian_window(sigma, max_width)                      Source.cpp:60-73 */
int blur_gaussian_window(double sigma, int max_width);

/* getGaussian(kernel, sigma, width, FFT_length)          Source.cpp:75-102
   kernel must hold max(width, fft_length) floats (width==0 -> gaussian_window(sigma)). */
int blur_get_gaussian(float* kernel, double sigma, int width, int fft_length);

/* isValidSize / nearestTransformSize                     Utils.hpp:141-157 */
int blur_is_valid_size(int n);
int blur_nearest_transform_size(int n);

/* the sizing block of pffft_()                           Source.cpp:434-457
   out = { kSize, pad, sizes[0] (column FFT length), sizes[1] (row FFT length),
           trailing_zeros[0], trailing_zeros[1] } */
int blur_pffft_sizing(int rows, int cols, double sigma, int out[6]);

/* per-bin multiplier kerf[2i]*scaler of Source.cpp:423 for bins 0..n/2 (n/2+1 floats);
   the kernel spectrum is computed on the host in float64 and rounded to float once */
int blur_kernel_multipliers(double sigma, int ksize, int n, float* m);

/* radix sequence the engine uses for a complex FFT of length n (returns the number of
   passes, 0 if n is unsupported); radices[] must hold 16 ints */
int blur_fft_plan_radices(int n, int* radices);

/* ---- context ---------------------------------------------------------------------- */

/* device: HIP device ordinal.  Owns plan/twiddle caches, kernel-spectrum caches and the
   float32 intermediate workspace.  (Role of pffft_new_setup/pffft_destroy_setup,
   Source.cpp:477-478,565-566, which the reference rebuilds on every call.) */
int blur_ctx_create(blur_ctx** out, int device);
int blur_ctx_destroy(blur_ctx* ctx);
/* hipStream_t to launch on (NULL = the default stream).  Switching to a different stream first
   waits for the work queued on the previous one (the workspace is shared). */
int blur_ctx_set_stream(blur_ctx* ctx, void* hip_stream);
int blur_ctx_synchronize(blur_ctx* ctx);
/* message of the last failing call on this ctx ("" if none); ctx may be NULL for
   failures of blur_ctx_create */
const char* blur_last_error(const blur_ctx* ctx);
/* which kernels the last u8c3 blur on this ctx ran on: returns 0 run-time-planned FFT, 1 specialised rows-first FFT, 2 wave-resident
   FFT, 3 whole-image 2D FFT, 4 two-kernel matrix-core engine, 6 fused matrix-core kernel (-1: none yet), and writes into note
   (n bytes, may be NULL) the engine's name and, under BLUR_ENGINE_AUTO, why a faster engine was passed over -- e.g.
   "two-kernel matrix-core engine (not taken: wide fused kernel (pad 73 .. 168): frames below 1 MP run on two kernels or the FFT kernels)" */
int blur_last_engine(const blur_ctx* ctx, char* note, size_t n);

/* Per-kernel timing with HIP events on the ctx's stream.  While enabled, every launch
   of the row-pass and column-pass kernels is bracketed by events; blur_ctx_timing()
   synchronises, then returns the summed milliseconds, the launch counts and the number of
   frames those launches covered (a batch launch processes several frames) since the last
   reset:  out_ms[0]=row pass, out_ms[1]=column pass; the others likewise.  (The fused engine: [0] = the fused kernel,
   [1] = its side kernels per call.)  on = 2 brackets slot 0 only: an event between two kernels keeps the second from
   starting while the first drains, about 3.5 us each on MI355X, and a measurement run may want to pay that once per call. */
int blur_ctx_timing_enable(blur_ctx* ctx, int on);
int blur_ctx_timing(blur_ctx* ctx, double out_ms[2], int out_launches[2], int out_frames[2], int reset);

/* ---- the hot path: pffft_(image, sigma)             Source.cpp:429-570 ------------- */

/* One BGR/RGB u8 frame, interleaved, rows*cols*3 bytes contiguous (cv::Mat::data of
   Source.cpp:459-461,567), DEVICE pointers.  dst may equal src (the reference works in
   place).  Asynchronous on the ctx's stream. */
int blur_gaussian_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst,
                           int rows, int cols, double sigma, const blur_opts* opts);

/* nframes frames of identical shape stored back to back (frame stride rows*cols*3). */
int blur_gaussian_u8c3_batch_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int nframes,
                                 int rows, int cols, double sigma, const blur_opts* opts);

/* One float32 plane (the per-channel body Source.cpp:510-564; BASELINE config 1). */
int blur_gaussian_f32c1_dev(blur_ctx* ctx, const float* d_src, float* d_dst,
                            int rows, int cols, double sigma, const blur_opts* opts);

/* HOST pointers: copy in, blur, copy out, synchronise (what a cv::Mat caller needs). */
int blur_gaussian_u8c3_host(blur_ctx* ctx, const uint8_t* src, uint8_t* dst,
                            int rows, int cols, double sigma, const blur_opts* opts);
int blur_gaussian_f32c1_host(blur_ctx* ctx, const float* src, float* dst,
                             int rows, int cols, double sigma, const blur_opts* opts);
/* nframes frames back to back in host memory (a video-style caller: the loop around pffft_() in Test(),
   Source.cpp:627-635, with the frames of a clip instead of sigmas).  Frame i+1 is copied to the device and frame i-1
   back while frame i is in the kernels; with pinned host memory (blur_host_alloc) the copies run at PCIe rate in both
   directions at once, with pageable memory the call is still correct but the copies serialise.  src == dst allowed.
   Synchronous. */
int blur_gaussian_u8c3_host_batch(blur_ctx* ctx, const uint8_t* src, uint8_t* dst, int nframes,
                                  int rows, int cols, double sigma, const blur_opts* opts);
/* the same for images whose rows are src_pitch / dst_pitch BYTES apart (cv::Mat::step of a ROI or
   of a padded Mat; pffft_() itself assumes image.data is continuous, Source.cpp:459-461) */
int blur_gaussian_u8c3_host_pitched(blur_ctx* ctx, const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch,
                                    int rows, int cols, double sigma, const blur_opts* opts);

/* Row pass only (Source.cpp:520-537): u8c3 frame -> three float planes, row-major
   (what `resf` holds at Source.cpp:536).  d_planes: 3*rows*cols floats.  For tests. */
int blur_rowpass_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, float* d_planes,
                          int rows, int cols, double sigma, const blur_opts* opts);

/* ---- other separable kernels on the same engine ------------------ */

/* Any symmetric separable kernel instead of getGaussian(): taps[ksize] (HOST pointer, odd ksize,
   centre tap in the middle, taps[i] == taps[ksize-1-i] so the spectrum is real, Source.cpp:419),
   reflect-101 padding of `pad` pixels (>= ksize/2 for a linear convolution); the FFT lengths
   follow Source.cpp:445-457 from pad.  Device frame pointers, asynchronous. */
int blur_separable_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols,
                            const float* taps, int ksize, int pad, const blur_opts* opts);

/* The `#define boxblur` mode of pffft_() (Source.cpp:437-442,468-472): FFT-domain tent kernel
   box_kernel(nsmooth^2) with passes = 2 padding. */
int blur_boxfft_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols,
                         double nsmooth, const blur_opts* opts);
/* its sizing: out = { kLen, pad, sizes[0], sizes[1] }                     Source.cpp:437-457 */
int blur_boxfft_sizing(int rows, int cols, double nsmooth, int out[4]);
/* box_kernel(kernel, kLen, FFT_length), 1D form: fft_length floats (host)  Source.cpp:129-140 */
int blur_box_kernel(float* kernel, int klen, int fft_length);

/* ---- pocketfft_2D: the whole padded image as ONE 2D transform      Source.cpp:143-277 ---- */

/* its sizing: out = { kSize, pad, sizes[0], sizes[1], border top, bottom, left, right }   Source.cpp:149-176
   (extra padding for a 2^a 3^b 5^c side goes into the reflect-101 borders of that axis: floor before, ceil after) */
int blur_pocketfft2d_sizing(int rows, int cols, double sigma, int out[8]);
/* pocketfft_2D(image, sigma): Reflect_101 on four sides, deinterleave, r2c over both axes, multiply by
   Re(kerf_1D_row[j]) Re(kerf_1D_col[i]), c2r with 1/ndata, interleave ("+0.5f, truncate"), crop   (:178-276).
   dft_image != 0 builds the `#define DFT_image` variant instead (:235-252): every plane of the result is the
   fft-shifted log spectrum 20 log10(|Re F| + 1e-5) of the padded plane, read with the reference's index arithmetic,
   then interleaved and cropped like the blur.  d_planes (optional, may be NULL): the cropped float planes
   [3][rows][cols] before the "+0.5f, truncate".  d_dst may equal d_src.  Asynchronous on the context's stream.
   In exact arithmetic the blur equals blur_gaussian_u8c3_dev with nyquist_quirk = 0, which is several times faster
   (two fused kernels instead of six); this entry point exists for the spectrum image and for callers who want the
   reference's 2D structure (its sizes and borders) reproduced step by step.
   BLUR_ERR_UNSUPPORTED when a border exceeds dim - 1 (Reflect_101 clamps there, Utils.hpp:217-220, and the
   reference's own buffers stop agreeing) or a side does not fit the LDS (about 19000). */
int blur_pocketfft2d_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols, double sigma,
                              int dft_image, float* d_planes);
int blur_pocketfft2d_u8c3_host(blur_ctx* ctx, const uint8_t* src, uint8_t* dst, int rows, int cols, double sigma, int dft_image);

/* ---- the pieces either side of it -------------------------------------------------- */

/* Reflect_101<uint8_t, C>(input, output, top, bottom, left, right, original_size)       Utils.hpp:212-243
   borders are clamped to dim - 1 like the reference; out_size = { rows + top + bottom, cols + left + right } after the
   clamp (d_out == NULL: size query only).  Out of place. */
int blur_reflect101_u8_dev(blur_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, int rows, int cols, int channels,
                           int top, int bottom, int left, int right, int out_size[2]);

/* flip_block<float,1>(in, out, w, h): out[x*h+y] = in[y*w+x]     call sites Source.cpp:540,562 */
int blur_flip_block_f32_dev(blur_ctx* ctx, const float* d_in, float* d_out, int w, int h);

/* deinterleave_BGR<uint8_t,float> / interleave_BGR<uint8_t,float>   Utils.hpp:159-210
   d_planes: 3 planes of `total` floats, back to back. */
int blur_deinterleave_bgr_u8_f32_dev(blur_ctx* ctx, const uint8_t* d_in, float* d_planes, uint32_t total);
int blur_interleave_bgr_f32_u8_dev(blur_ctx* ctx, const float* d_planes, uint8_t* d_out, uint32_t total);

/* fastboxblur(in, w, h, channels, ksize, passes), in place     call site Source.cpp:587 */
int blur_fastboxblur_u8_dev(blur_ctx* ctx, uint8_t* d_inout, int w, int h, int channels,
                            int ksize, int passes);
int blur_fastboxblur_u8_host(blur_ctx* ctx, uint8_t* inout, int w, int h, int channels,
                             int ksize, int passes);

/* ---- several GPUs behind one handle (SURVEY.md 8(b) S1 "ctx owning a device list", 8(e)) -------------------------------
   Frames are independent (Source.cpp:510 runs even the channels serially; the reference's only parallelism is
   hybrid_loop over tiles, Utils.hpp:16-55), so a batch shards by frame with no exchange: shard r of S takes frames
   [n r / S, n (r + 1) / S), has its own context (plans and kernel spectra are deterministic host code: nothing to
   broadcast) and its own stream; one host thread drives all of them.  `devices` may repeat an ordinal (several logical
   shards on one GPU).  Both calls are SYNCHRONOUS.
     ..._multi_dev : frames in the memory of devices[0]; shards on other GPUs receive and return their frames by peer
                     copies over xGMI (point to point: no ring, nothing to reduce), shards on devices[0] work in place;
     ..._multi_host: frames in host memory (blur_host_alloc for DMA without staging); every shard copies its own slice. */
typedef struct blur_multi blur_multi;
int blur_multi_create(blur_multi** out, const int* devices, int ndevices);
int blur_multi_destroy(blur_multi* m);
int blur_multi_shards(const blur_multi* m);
const char* blur_multi_last_error(const blur_multi* m);
int blur_gaussian_u8c3_batch_multi_dev(blur_multi* m, const uint8_t* d_src, uint8_t* d_dst, int nframes, int rows, int cols,
                                       double sigma, const blur_opts* opts);
int blur_gaussian_u8c3_batch_multi_host(blur_multi* m, const uint8_t* src, uint8_t* dst, int nframes, int rows, int cols,
                                        double sigma, const blur_opts* opts);

/* ---- batched line convolution: what pffft_transform_ordered(FORWARD) -> pffft_sorted_optimized_convolution ->
   pffft_transform_ordered(BACKWARD) (Source.cpp:531-533,553-555) is per tile, for MANY lines at once --------------
   d_in / d_out: nlines complex lines of n points each (interleaved re, im floats; in == out allowed),
   out = IDFT_n(multipliers .* DFT_n(in)), UNNORMALISED like pffft (fold 1/n into the multipliers, Source.cpp:423);
   multipliers: n real factors in natural frequency order (host pointer; cached on the device by content).
   Two real lines ride in one complex line when the multipliers are even (m[f] = m[n-f]): re and im are then
   convolved independently.  n must be a length the wave-resident kernels support: blur_wr_length(). */
int blur_convolve_lines_c32_dev(blur_ctx* ctx, const float* d_in, float* d_out, int nlines, int n, const float* multipliers);
/* ---- matrix-core engine (mx_kernels.hpp): both passes as banded Toeplitz products on v_mfma_f32_32x32x16_f16 ----
   blur_opts.engine = BLUR_ENGINE_MATRIX selects it for the u8c3 entry points (BLUR_ENGINE_FUSED: the one-kernel form of the same
   products, fx_kernels.hpp, which is the library's own choice where it applies).  Same linear map as the FFT product inside the
   crop (Source.cpp:536,558), Nyquist-slot quirk (Source.cpp:420-425) included as a rank-one term per line.
   blur_mx_window_blocks(pad): window blocks (of 16 positions) of the kernel instantiated for this pad, 0 = none.
   blur_mx_fragments: the Toeplitz operand fragments the kernels load, [2][nkb][64][8] binary16 (hi, lo halves of
   taps * 2^14); taps: 2 pad + 1 floats, centre at index pad (host only; tests). */
int blur_mx_window_blocks(int pad);
int blur_mx_fragments(const float* taps, int pad, int nkb, uint16_t* out);

/* smallest supported transform length >= need for the column (1) or row (0) role; 0 if there is none.
   (The engine's transform length need not be nearestTransformSize(): only the Nyquist-slot term of Source.cpp:420-425
   depends on the reference's length, and the multiplier of bin n/2 reproduces it.) */
int blur_wr_length(int need, int column_role);
/* the multipliers those kernels use for the Gaussian, all n bins in natural order: m[f] = Re DFT_n(kernel)[f] / n, and with
   quirk != 0 bin n/2 carries the reference's Nyquist-slot term for ITS transform length n_ref (Source.cpp:420-425).  Host only. */
int blur_wr_kernel_multipliers(double sigma, int ksize, int n, int n_ref, int quirk, float* m);

/* ---- plain device-memory plumbing for callers without a HIP runtime of their own --- */
int blur_malloc(blur_ctx* ctx, void** d_ptr, size_t bytes);
int blur_free(blur_ctx* ctx, void* d_ptr);
int blur_memcpy_h2d(blur_ctx* ctx, void* d_dst, const void* src, size_t bytes);   /* synchronous */
int blur_memcpy_d2h(blur_ctx* ctx, void* dst, const void* d_src, size_t bytes);   /* synchronous */
/* page-locked host memory (the role PFAlloc plays for pffft's aligned buffers, Utils.hpp:57-138: memory the
   transport wants): host images that live here are copied by DMA without a staging pass */
int blur_host_alloc(blur_ctx* ctx, void** h_ptr, size_t bytes);
int blur_host_free(blur_ctx* ctx, void* h_ptr);

/* ---- measurement: the box's streaming rate as THIS library's kernels would see it: a 16-byte-per-lane copy of `bytes` bytes
   (device to device, `reps` launches between two events on the context's stream); *gbs = (bytes read + bytes written) / s / 1e9.
   bench.py prints it beside the 8 TB/s spec figure (MI355X_MICROARCH.md gives 6.29 TB/s for this access shape). */
int blur_copy_bandwidth(blur_ctx* ctx, size_t bytes, int reps, double* gbs);

#ifdef __cplusplus
}
#endif
#endif /* BLUR_AMD_H */
