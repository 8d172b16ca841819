// engine.hip -- HIP kernels (gfx950) and the C ABI of include/blur_amd.h.
//
// Data path of pffft_() (Source.cpp:429-570) on the GPU, per frame:
//
//   rowpass_kernel   u8 BGR interleaved (or one f32 plane)  --->  3 float planes, row-major
//       deinterleave_BGR (Utils.hpp:159-184) + per-row reflect-101 pad (Source.cpp:525-529)
//       + FFT / pointwise / inverse FFT (:531-533) + crop (:536), two image rows per
//       complex line, one workgroup per (row pair, channel).
//   colpass_kernel   3 float planes  --->  u8 BGR interleaved (or one f32 plane)
//       a workgroup owns a strip of G adjacent columns and reads it straight out of the
//       row-major planes (the transposes flip_block of :540,562 never materialise), pads
//       by reflection along the column (:549-551), FFT / pointwise / inverse FFT (:553-555),
//       crops (:558) and applies interleave_BGR's "+0.5f, truncate" (Utils.hpp:189,204-206),
//       staging the three channels in LDS so that the strip is written as whole pixels.
//
// HBM traffic per pixel: 3 B in + 12 B intermediate out, 12 B intermediate in + 3 B out.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/blur_amd.h"
#include "fast_registry.hpp"
#include "fft_engine.hpp"
#include "host_math.hpp"
#define BLUR_MX_QUIRK_KERNELS
#include "mx_registry.hpp"
#define BLUR_FX_QUIRK_KERNELS
#include "fx_registry.hpp"
#include "bx_box.hpp"
#include "wr_registry.hpp"

using namespace blur_amd;

namespace {

constexpr int kThreads = 256;
constexpr size_t kLdsLimit = 160 * 1024;

// Blocks b and b + nx share an XCD (and its L2; nx = hipDeviceAttributeNumberOfXccs, 8 on MI355X).  Give each XCD a contiguous
// run of work items so that neighbours (which touch the same 128-byte lines) meet in one L2.
__device__ __forceinline__ int xcd_contiguous(int b, int nwg, int nx)
{
    if (nx < 1) nx = 8;
    const int q = nwg / nx, r = nwg - q * nx, x = b % nx;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / nx;
}

// reflect-101 source index of padded position p (Source.cpp:525-529); -1 = trailing zero
__device__ __forceinline__ int reflect_src(int p, int pad, int len)
{
    const int i = p - pad;
    if (i < 0) return -i;
    if (i < len) return i;
    if (i < len + pad) return 2 * (len - 1) - i;
    return -1;
}

template <typename T> __device__ __forceinline__ float load_px(const T* p) { return static_cast<float>(*p); }

// ------------------------------------------------------------------------------------
// row pass: one workgroup = rows (2q, 2q+1) of channel c
// ------------------------------------------------------------------------------------
template <typename InT, int CH, int T = kThreads>
__global__ __launch_bounds__(T) void rowpass_kernel(const InT* __restrict__ src, float* __restrict__ planes,
                                                           int rows, int cols, int pad, DevPlan plan,
                                                           const float2* __restrict__ tw, const float* __restrict__ mperm, int gshift)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* z = reinterpret_cast<float2*>(smem);
    const int item = xcd_contiguous(blockIdx.x, gridDim.x, plan.nxcd);
    const int pair = item / CH, c = item - pair * CH;
    const int r0 = 2 * pair;
    const bool two = r0 + 1 < rows;
    const int n = plan.n;
    const InT* row_a = src + (static_cast<size_t>(r0) * cols) * CH + c;
    const InT* row_b = src + (static_cast<size_t>(r0 + (two ? 1 : 0)) * cols) * CH + c;

    // unconditional loads (clamped index, value masked afterwards): a load under a branch costs one
    // memory round trip per iteration, unconditional ones overlap across the unrolled iterations
#pragma unroll 8
    for (int p = threadIdx.x; p < n; p += T) {
        const int x = reflect_src(p, pad, cols);
        const size_t xi = static_cast<size_t>(x >= 0 ? x : 0) * CH;
        const float a = load_px(row_a + xi), b = load_px(row_b + xi);
        z[phys(p)] = make_float2(x >= 0 ? a : 0.f, (x >= 0 && two) ? b : 0.f);
    }
    __syncthreads();
    fftconv_lines<1>(z, 0, plan, tw, mperm);

    if (gshift > 0) {
        // strip-major intermediate (gshift = log2 of the G columns a column-pass workgroup takes): [channel][strip x / G][row][G], so
        // that the column pass reads its strip as ONE contiguous block instead of 4 G bytes out of every row of the plane
        const int G = 1 << gshift, nstrips = (cols + G - 1) >> gshift;
        float* base = planes + static_cast<size_t>(c) * nstrips * rows * G + static_cast<size_t>(r0) * G;
        for (int x = threadIdx.x; x < cols; x += T) {
            const float2 v = z[phys(pad + x)];
            float* o = base + static_cast<size_t>(x >> gshift) * rows * G + (x & (G - 1));
            o[0] = v.x;
            if (two) o[G] = v.y;
        }
        return;
    }
    float* out_a = planes + (static_cast<size_t>(c) * rows + r0) * cols;
    float* out_b = out_a + cols;
    for (int x = threadIdx.x; x < cols; x += T) {
        const float2 v = z[phys(pad + x)];
        out_a[x] = v.x;
        if (two) out_b[x] = v.y;
    }
}

// ------------------------------------------------------------------------------------
// column pass: one workgroup = columns [x0, x0 + 2C) of all CH channels
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void store_out(float* p, float v) { *p = v; }

template <typename OutT, int CH, int C, int T = kThreads>
__global__ __launch_bounds__(T) void colpass_kernel(const float* __restrict__ planes, OutT* __restrict__ dst,
                                                           int rows, int cols, int pad, DevPlan plan,
                                                           const float2* __restrict__ tw, const float* __restrict__ mperm, int strips)
{
    constexpr int G = 2 * C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* z = reinterpret_cast<float2*>(smem);
    const int n = plan.n;
    const int zs = line_stride(n);
    uint8_t* stage = reinterpret_cast<uint8_t*>(z + static_cast<size_t>(C) * zs);   // [rows][G][CH] bytes (u8 output only)
    const int strip = xcd_contiguous(blockIdx.x, gridDim.x, plan.nxcd);
    const int x0 = strip * G;

    for (int c = 0; c < CH; ++c) {
        // strips: the intermediate is [channel][strip][row][G] (rowpass_kernel), this workgroup's strip one contiguous block
        const int nstrips = (cols + G - 1) / G;
        const float* plane = strips ? planes + (static_cast<size_t>(c) * nstrips + strip) * rows * G : planes + static_cast<size_t>(c) * rows * cols;
        // gather the strip: position p of line l  <-  plane[reflect(p)][x0 + 2l, x0 + 2l + 1]
#pragma unroll 4
        for (int idx = threadIdx.x; idx < n * C; idx += T) {
            const int p = idx / C, l = idx - p * C;
            const int r = reflect_src(p, pad, rows);
            const int col = x0 + 2 * l;
            const float* s = strips ? plane + static_cast<size_t>(r >= 0 ? r : 0) * G - x0 : plane + static_cast<size_t>(r >= 0 ? r : 0) * cols;
            const float a = strips ? s[col] : s[col < cols ? col : cols - 1], b = strips ? s[col + 1] : s[col + 1 < cols ? col + 1 : cols - 1];   // unconditional
            z[l * zs + phys(p)] = make_float2((r >= 0 && col < cols) ? a : 0.f, (r >= 0 && col + 1 < cols) ? b : 0.f);
        }
        __syncthreads();
        fftconv_lines<C>(z, zs, plan, tw, mperm);

        if constexpr (sizeof(OutT) == 1) {
            // interleave_BGR<uint8_t,float>: (uint8_t)(value + 0.5f)   Utils.hpp:189,204-206
            for (int idx = threadIdx.x; idx < rows * C; idx += T) {
                const int r = idx / C, l = idx - r * C;
                const float2 v = z[l * zs + phys(pad + r)];
                uint8_t* s = stage + (static_cast<size_t>(r) * G + 2 * l) * CH + c;
                s[0] = static_cast<uint8_t>(static_cast<int>(v.x + 0.5f));
                s[CH] = static_cast<uint8_t>(static_cast<int>(v.y + 0.5f));
            }
        } else {
            for (int idx = threadIdx.x; idx < rows * C; idx += T) {
                const int r = idx / C, l = idx - r * C;
                const float2 v = z[l * zs + phys(pad + r)];
                const int col = x0 + 2 * l;
                OutT* d = dst + (static_cast<size_t>(r) * cols + col) * CH + c;
                if (col < cols) d[0] = v.x;
                if (col + 1 < cols) d[CH] = v.y;
            }
        }
        __syncthreads();
    }

    if constexpr (sizeof(OutT) == 1) {
        const int wbytes = (min(G, cols - x0)) * CH;      // valid bytes per row of the strip
        for (int idx = threadIdx.x; idx < rows * G * CH; idx += T) {
            const int r = idx / (G * CH), b = idx - r * (G * CH);
            if (b < wbytes) dst[(static_cast<size_t>(r) * cols + x0) * CH + b] = stage[idx];
        }
    }
}

// ------------------------------------------------------------------------------------
// the pieces: flip_block, de/interleave
// ------------------------------------------------------------------------------------
// flip_block<float,1>: out[x*h + y] = in[y*w + x]; 64x64 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void flip_block_kernel(const float* __restrict__ in, float* __restrict__ out, int w, int h)
{
    __shared__ float tile[64][65];
    const int tiles_x = (w + 63) / 64;
    const int t = blockIdx.x;
    const int tx = (t % tiles_x) * 64, ty = (t / tiles_x) * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int yy = ly; yy < 64; yy += 4) {
        const int x = tx + lx, y = ty + yy;
        if (x < w && y < h) tile[yy][lx] = in[static_cast<size_t>(y) * w + x];
    }
    __syncthreads();
    for (int xx = ly; xx < 64; xx += 4) {
        const int x = tx + xx, y = ty + lx;
        if (x < w && y < h) out[static_cast<size_t>(x) * h + y] = tile[lx][xx];
    }
}

__global__ __launch_bounds__(256) void deinterleave_kernel(const uint8_t* __restrict__ in, float* __restrict__ planes, uint32_t total)
{
    for (uint32_t x = blockIdx.x * 256u + threadIdx.x; x < total; x += gridDim.x * 256u) {
        planes[x] = in[3 * static_cast<size_t>(x)];
        planes[static_cast<size_t>(total) + x] = in[3 * static_cast<size_t>(x) + 1];
        planes[2 * static_cast<size_t>(total) + x] = in[3 * static_cast<size_t>(x) + 2];
    }
}

__global__ __launch_bounds__(256) void interleave_kernel(const float* __restrict__ planes, uint8_t* __restrict__ out, uint32_t total)
{
    for (uint32_t x = blockIdx.x * 256u + threadIdx.x; x < total; x += gridDim.x * 256u) {
        out[3 * static_cast<size_t>(x)] = static_cast<uint8_t>(static_cast<int>(planes[x] + 0.5f));
        out[3 * static_cast<size_t>(x) + 1] = static_cast<uint8_t>(static_cast<int>(planes[static_cast<size_t>(total) + x] + 0.5f));
        out[3 * static_cast<size_t>(x) + 2] = static_cast<uint8_t>(static_cast<int>(planes[2 * static_cast<size_t>(total) + x] + 0.5f));
    }
}

// ------------------------------------------------------------------------------------
// tiled wave-resident path (run_wr_tiled): the Nyquist-slot quirk's two term vectors from fx_prepass's integer sums
//   e[f][c][x] = dc Ccol(x, c)                                                (x < pitch; 0 right of the image)
//   h[f][c][r] = dr (colconv(Srow)(r, c) + dc (-1)^(r + pad) Z(c))            (the column convolution with reflect-101, in double)
// grid (ne + nh blocks of 256 outputs, frames): the first ne = ceil(pitch / 256) blocks make e, the others h
// dynamic LDS: 3 (256 + 2 pad) + 2 pad + 1 + 3 + 3 x 256 doubles
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tl_terms_kernel(const int* __restrict__ srow_part, const int* __restrict__ cpart, const long long* __restrict__ zpart,
                                                       const float* __restrict__ taps, float* __restrict__ e, float* __restrict__ h, int rows, int cols, int pad,
                                                       int pitch, int nbatches, int nbands, int cpitch, float dr, float dc, int ne)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tl_lds[];
    const int f = blockIdx.y, tid = threadIdx.x;
    if (static_cast<int>(blockIdx.x) < ne) {
        const int x = blockIdx.x * 256 + tid;
        if (x >= pitch) return;
        // the bands' parts in batches of eight independent loads per channel (a band's part is below 2^31 / nbands: exact in int)
        const int* cp = cpart + static_cast<size_t>(f) * nbands * cpitch + 3 * (x < cols ? x : 0);
        long long tot[3] = { 0, 0, 0 };
        for (int b0 = 0; b0 < nbands; b0 += 8) {
            int t[8][3];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) t[j][c] = cp[static_cast<size_t>(b0 + j < nbands ? b0 + j : nbands - 1) * cpitch + c];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) tot[c] += b0 + j < nbands ? t[j][c] : 0;
        }
        for (int c = 0; c < 3; ++c)
            e[(static_cast<size_t>(f) * 3 + c) * pitch + x] = x < cols ? static_cast<float>(static_cast<double>(dc) * static_cast<double>(tot[c])) : 0.f;
        return;
    }
    const int r0 = (blockIdx.x - ne) * 256, win = 256 + 2 * pad, ntap = 2 * pad + 1;
    double* sr = reinterpret_cast<double*>(tl_lds);            // [3][win]
    double* tp = sr + 3 * win;
    double* zs = tp + ntap;                                    // [3]
    double* zr = zs + 3;                                       // [3][256] partial sums of Z
    for (int i = tid; i < 3 * win; i += 256) {
        const int c = i / win, p = i - c * win;
        const int r = mx_refl(r0 - pad + p, rows);
        long long v = 0;
        for (int b = 0; b < nbatches; ++b) v += srow_part[((static_cast<size_t>(f) * nbatches + b) * rows + r) * 3 + c];
        sr[i] = static_cast<double>(v);
    }
    for (int i = tid; i < ntap; i += 256) tp[i] = static_cast<double>(taps[i]);
    {
        const int nz = nbands * nbatches;
        long long z[3] = { 0, 0, 0 };
        for (int i = tid; i < nz; i += 256)
            for (int c = 0; c < 3; ++c) z[c] += zpart[(static_cast<size_t>(f) * nz + i) * 3 + c];
        for (int c = 0; c < 3; ++c) zr[c * 256 + tid] = static_cast<double>(z[c]);      // (exact: |Z| < 2^53)
    }
    __syncthreads();
    if (tid < 3) {
        double z = 0;
        for (int i = 0; i < 256; ++i) z += zr[tid * 256 + i];
        zs[tid] = z;
    }
    __syncthreads();
    const int r = r0 + tid;
    if (r >= rows) return;
    const double sg = ((r + pad) & 1) ? -1.0 : 1.0;
    for (int c = 0; c < 3; ++c) {
        const double* s = sr + c * win + tid;
        double acc[4] = { 0, 0, 0, 0 };
        int t = 0;
        for (; t + 4 <= ntap; t += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fma(tp[t + j], s[t + j], acc[j]);
        }
        for (; t < ntap; ++t) acc[0] = __builtin_fma(tp[t], s[t], acc[0]);
        h[(static_cast<size_t>(f) * 3 + c) * rows + r] =
            static_cast<float>(static_cast<double>(dr) * (((acc[0] + acc[1]) + (acc[2] + acc[3])) + static_cast<double>(dc) * sg * zs[c]));
    }
}

// ------------------------------------------------------------------------------------
// fastboxblur: one sweep of the sliding accumulator along lines of length n.
// Element (line l, position x, channel c) at buf[l*lstride + x*xstride + c].
// A wave owns 64 adjacent (line, channel) accumulators; along the line it walks
// sequentially (the accumulator recurrence), so for the vertical sweep (xstride = row
// pitch) the 64 lanes read 64 adjacent bytes, and for the horizontal sweep the lines are
// first brought through LDS in chunks.  V1: direct global accesses, L2 absorbs the reuse.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int refl101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void boxsweep_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int nlines, int n,
                                                       size_t lstride, size_t xstride, int C, int r, int seg_len)
{
    // work item = (segment, line*C + c); segments split long lines so the grid fills the chip
    const int lc_total = nlines * C;
    const int nseg = (n + seg_len - 1) / seg_len;
    const long long gid = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (gid >= static_cast<long long>(lc_total) * nseg) return;
    const int lc = static_cast<int>(gid % lc_total), seg = static_cast<int>(gid / lc_total);
    const int l = lc / C, c = lc - l * C;
    const uint8_t* ip = in + static_cast<size_t>(l) * lstride + c;
    uint8_t* op = out + static_cast<size_t>(l) * lstride + c;
    const float iarr = 1.f / static_cast<float>(r + r + 1);
    const int xs = seg * seg_len, xe = min(n, xs + seg_len);
    int acc = 0;
    for (int d = -r; d <= r; ++d) acc += ip[static_cast<size_t>(refl101(xs + d, n)) * xstride];
    op[static_cast<size_t>(xs) * xstride] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
    for (int x = xs + 1; x < xe; ++x) {
        acc += ip[static_cast<size_t>(refl101(x + r, n)) * xstride];
        acc -= ip[static_cast<size_t>(refl101(x - r - 1, n)) * xstride];
        op[static_cast<size_t>(x) * xstride] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
    }
}

// Vertical sweep, four adjacent bytes (column-channels) per thread: dword loads/stores, lanes
// cover 256 contiguous bytes of an image row, the walk along y is the accumulator recurrence.
// Segments along y give the grid enough threads; each pays 2r+1 loads to start its window.
__global__ __launch_bounds__(256) void boxcol4_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      int h, int pitch4 /* dwords per row */, int r, int seg_len, int nseg)
{
    const long long gid = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (gid >= static_cast<long long>(pitch4) * nseg) return;
    const int xq = static_cast<int>(gid % pitch4), seg = static_cast<int>(gid / pitch4);
    const uint32_t* ip = reinterpret_cast<const uint32_t*>(in) + xq;
    uint32_t* op = reinterpret_cast<uint32_t*>(out) + xq;
    const float iarr = 1.f / static_cast<float>(r + r + 1);
    const int ys = seg * seg_len, ye = min(h, ys + seg_len);
    if (ys >= ye) return;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    auto add = [&](int y, int sgn) {
        const uint32_t v = ip[static_cast<size_t>(refl101(y, h)) * pitch4];
        a0 += sgn * static_cast<int>(v & 0xffu);
        a1 += sgn * static_cast<int>((v >> 8) & 0xffu);
        a2 += sgn * static_cast<int>((v >> 16) & 0xffu);
        a3 += sgn * static_cast<int>(v >> 24);
    };
    auto emit = [&](int y) {
        const uint32_t b0 = static_cast<uint32_t>(static_cast<int>(static_cast<float>(a0) * iarr + 0.5f)) & 0xffu;
        const uint32_t b1 = static_cast<uint32_t>(static_cast<int>(static_cast<float>(a1) * iarr + 0.5f)) & 0xffu;
        const uint32_t b2 = static_cast<uint32_t>(static_cast<int>(static_cast<float>(a2) * iarr + 0.5f)) & 0xffu;
        const uint32_t b3 = static_cast<uint32_t>(static_cast<int>(static_cast<float>(a3) * iarr + 0.5f)) & 0xffu;
        op[static_cast<size_t>(y) * pitch4] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    };
    for (int d = -r; d <= r; ++d) add(ys + d, 1);
    emit(ys);
    int y = ys + 1;
    if (ys - r - 1 >= 0 && ye + r < h) {                // interior segment: loads of four rows in flight together
        for (; y + 4 <= ye; y += 4) {
            uint32_t vi[4], vo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vi[i] = ip[static_cast<size_t>(y + i + r) * pitch4];
                vo[i] = ip[static_cast<size_t>(y + i - r - 1) * pitch4];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0 += static_cast<int>(vi[i] & 0xffu) - static_cast<int>(vo[i] & 0xffu);
                a1 += static_cast<int>((vi[i] >> 8) & 0xffu) - static_cast<int>((vo[i] >> 8) & 0xffu);
                a2 += static_cast<int>((vi[i] >> 16) & 0xffu) - static_cast<int>((vo[i] >> 16) & 0xffu);
                a3 += static_cast<int>(vi[i] >> 24) - static_cast<int>(vo[i] >> 24);
                emit(y + i);
            }
        }
    }
    for (; y < ye; ++y) {
        add(y + r, 1);
        add(y - r - 1, -1);
        emit(y);
    }
}

// All `passes` horizontal sweeps of one image row in one go: the row lives in LDS (two byte
// buffers, ping-pong), each thread walks one (channel, segment) with the sliding accumulator,
// u8 rounding after every sweep exactly as the per-sweep kernel does.  HBM sees the row once
// in and once out instead of once per pass.
__global__ __launch_bounds__(256) void boxrow_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                     int w, int C, int r, int passes, int seg_len, int nseg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rb = w * C;                               // bytes per row
    const int rb4 = (rb + 3) & ~3;
    uint8_t* a = reinterpret_cast<uint8_t*>(smem);
    uint8_t* b = a + rb4;
    const uint8_t* src = in + static_cast<size_t>(blockIdx.x) * rb;
    uint8_t* dst = out + static_cast<size_t>(blockIdx.x) * rb;
    if ((rb & 3) == 0) {
        for (int i = threadIdx.x; i < rb / 4; i += 256) reinterpret_cast<uint32_t*>(a)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < rb; i += 256) a[i] = src[i];
    }
    __syncthreads();
    const float iarr = 1.f / static_cast<float>(r + r + 1);
    for (int p = 0; p < passes; ++p) {
        for (int item = threadIdx.x; item < C * nseg; item += 256) {
            const int c = item % C, seg = item / C;
            const int xs = seg * seg_len, xe = min(w, xs + seg_len);
            if (xs >= xe) continue;
            // restrict: the sweep reads one buffer and writes the other, so the LDS reads of several steps may be
            // issued together instead of one read latency per step behind the previous step's store
            const uint8_t* __restrict__ s = a + c;
            uint8_t* __restrict__ d = b + c;
            int acc = 0;
            if (xs - r - 1 >= 0 && xe + r < w) {        // interior segment: no reflection anywhere
#pragma unroll 8
                for (int x = xs - r; x <= xs + r; ++x) acc += s[x * C];
                d[xs * C] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
                int x = xs + 1;
                for (; x + 8 <= xe; x += 8) {
                    int in_[8], out_[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { in_[i] = s[(x + i + r) * C]; out_[i] = s[(x + i - r - 1) * C]; }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc += in_[i] - out_[i];
                        d[(x + i) * C] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
                    }
                }
                for (; x < xe; ++x) {
                    acc += s[(x + r) * C] - s[(x - r - 1) * C];
                    d[x * C] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
                }
            } else {
                for (int x = xs - r; x <= xs + r; ++x) acc += s[refl101(x, w) * C];
                d[xs * C] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
                for (int x = xs + 1; x < xe; ++x) {
                    acc += s[refl101(x + r, w) * C] - s[refl101(x - r - 1, w) * C];
                    d[x * C] = static_cast<uint8_t>(static_cast<int>(static_cast<float>(acc) * iarr + 0.5f));
                }
            }
        }
        __syncthreads();
        uint8_t* t = a; a = b; b = t;
    }
    if ((rb & 3) == 0) {
        for (int i = threadIdx.x; i < rb / 4; i += 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(a)[i];
    } else {
        for (int i = threadIdx.x; i < rb; i += 256) dst[i] = a[i];
    }
}

// All `passes` horizontal sweeps of one image row, four pixels per step.  The row sits in LDS with reflect-101
// borders written out physically (r + 1 pixels on each side, refreshed after every sweep), so every segment takes
// the same branch-free path: per group of four pixels the entering and the leaving 4*C bytes are fetched as aligned
// dwords (+ v_alignbyte with a shift that is uniform for the whole launch), the 4*C outputs leave as C dwords.
// Against the byte-wise kernel above: 3.3x fewer LDS instructions in the walk, 2.9x fewer in the window set-up, no
// slow border segments.  Same integer accumulators and the same float rounding, so the bytes are identical.
template <int C>
__global__ __launch_bounds__(256) void boxrow4_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      int w, int r, int passes, int groups_per_thread, int off0, int buf_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* a = reinterpret_cast<uint8_t*>(smem);
    uint8_t* b = a + buf_bytes;
    const int rb = w * C;
    const uint8_t* src = in + static_cast<size_t>(blockIdx.x) * rb;
    uint8_t* dst = out + static_cast<size_t>(blockIdx.x) * rb;
    if ((rb & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
        for (int i = threadIdx.x; i < rb / 4; i += 256) reinterpret_cast<uint32_t*>(a + off0)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < rb; i += 256) a[off0 + i] = src[i];
    }
    const float iarr = 1.f / static_cast<float>(r + r + 1);
    const int xs = static_cast<int>(threadIdx.x) * groups_per_thread * 4;
    const int ngroups = xs < w ? min(groups_per_thread, (w - xs + 3) / 4) : 0;
    // byte offsets (mod 4) of the two streams: uniform because xs is a multiple of 4
    const int sh_in = (r * C) & 3, sh_out = ((-(r + 1)) * C) & 3;
    for (int p = 0; p < passes; ++p) {
        __syncthreads();                                     // the row (or the previous sweep) is complete
        // reflect-101 borders: pixels -(r+1) .. -1 and w .. w+r (the outermost one on the left only ever cancels out)
        for (int i = threadIdx.x; i < 2 * (r + 1) * C; i += 256) {
            const int side = i / ((r + 1) * C), k = i - side * (r + 1) * C;
            const int px = k / C, c = k - px * C;
            const int x = side ? w + px : -(px + 1);
            int m = x < 0 ? -x : 2 * (w - 1) - x;
            m = m < 0 ? 0 : (m > w - 1 ? w - 1 : m);
            a[off0 + x * C + c] = a[off0 + m * C + c];
        }
        __syncthreads();
        if (ngroups > 0) {
            const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(a);
            uint32_t* __restrict__ d32 = reinterpret_cast<uint32_t*>(b);
            // aligned dword index of the byte where a stream's group starts
            auto stream = [&](int byte_off, int sh, uint32_t (&g)[C]) {
                const int q = (off0 + byte_off - sh) >> 2;
                uint32_t t[C + 1];
#pragma unroll
                for (int k = 0; k <= C; ++k) t[k] = s32[q + k];
#pragma unroll
                for (int k = 0; k < C; ++k) g[k] = __builtin_amdgcn_alignbyte(t[k + 1], t[k], sh);
            };
            auto byte_of = [](const uint32_t (&g)[C], int k) -> int { return static_cast<int>((g[k >> 2] >> (8 * (k & 3))) & 0xffu); };
            int acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0;
            // window centred on pixel xs - 1: pixels xs-1-r .. xs-1+r = 2r+1 pixels from the "leaving" stream position
            {
                const int np = 2 * r + 1;
                int px = 0;
                for (; px + 4 <= np; px += 4) {
                    uint32_t g[C];
                    stream((xs - 1 - r + px) * C, sh_out, g);
#pragma unroll
                    for (int k = 0; k < 4 * C; ++k) acc[k % C] += byte_of(g, k);
                }
                for (; px < np; ++px) {
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += a[off0 + (xs - 1 - r + px) * C + c];
                }
            }
            for (int gi = 0; gi < ngroups; ++gi) {
                const int x = xs + 4 * gi;
                uint32_t gin[C], gout[C], res[C];
                stream((x + r) * C, sh_in, gin);
                stream((x - r - 1) * C, sh_out, gout);
#pragma unroll
                for (int k = 0; k < C; ++k) res[k] = 0;
#pragma unroll
                for (int k = 0; k < 4 * C; ++k) {
                    acc[k % C] += byte_of(gin, k) - byte_of(gout, k);
                    const uint32_t v = static_cast<uint32_t>(static_cast<int>(static_cast<float>(acc[k % C]) * iarr + 0.5f)) & 0xffu;
                    res[k >> 2] |= v << (8 * (k & 3));
                }
#pragma unroll
                for (int k = 0; k < C; ++k) d32[((off0 + x * C) >> 2) + k] = res[k];
            }
        }
        uint8_t* t = a; a = b; b = t;
    }
    __syncthreads();
    if ((rb & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
        for (int i = threadIdx.x; i < rb / 4; i += 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(a + off0)[i];
    } else {
        for (int i = threadIdx.x; i < rb; i += 256) dst[i] = a[off0 + i];
    }
}


// ------------------------------------------------------------------------------------
// Whole-image 2D transform path: pocketfft_2D (Source.cpp:143-277) and its DFT_image branch (:235-252).
//
// The padded image (Reflect_101 on all four sides, :178-180) is transformed as ONE 2D FFT.  Two colour planes ride in
// one complex image z = a + i b (the kernel spectrum is real, so they never mix in the blur; for the spectrum image
// they are separated with F_a[k] = (Z[k] + conj Z[-k]) / 2).  Spectra stay in the position order of the plans along
// both axes; nothing is reordered in memory.
//
//   img2d_rows_fwd   u8 image -> z rows: reflect-101 + deinterleave + pack fused into the load, forward FFT along x
//   img2d_cols<true> strip of G columns: (x Krow[pos_x]) -> forward along y -> x Kcol[pos_y] -> inverse along y  (:255-260)
//   img2d_cols<false>                    forward along y only (DFT_image keeps the spectrum)
//   img2d_rows_inv   rows of the crop: inverse along x, "+0.5f, truncate", crop (:263-276) -> u8 image
//   img2d_logspec    fftshift + 20 log10(|Re F| + 1e-5) with the reference's index arithmetic (:238-251), crop
// ------------------------------------------------------------------------------------
struct Img2dGeom { int rows, cols, s0, s1, top, left; };

__global__ __launch_bounds__(256) void img2d_rows_fwd_kernel(const uint8_t* __restrict__ src, float2* __restrict__ z, Img2dGeom g, int ca, int cb,
                                                             DevPlan plan, const float2* __restrict__ tw)
{
    extern __shared__ float2 lds2d[];
    const int zs = line_stride(plan.n);
    const int y = blockIdx.x;
    const uint8_t* row = src + static_cast<size_t>(refl101(y - g.top, g.rows)) * g.cols * 3;
    for (int x = threadIdx.x; x < g.s1; x += blockDim.x) {
        const uint8_t* px = row + 3 * refl101(x - g.left, g.cols);
        lds2d[phys(x)] = make_float2(static_cast<float>(px[ca]), cb >= 0 ? static_cast<float>(px[cb]) : 0.f);
    }
    __syncthreads();
    fft_forward_lines<1>(lds2d, zs, plan, tw);
    float2* out = z + static_cast<size_t>(y) * g.s1;
    for (int pos = threadIdx.x; pos < g.s1; pos += blockDim.x) out[pos] = lds2d[phys(pos)];
}

template <int G, bool CONV>
__global__ __launch_bounds__(256) void img2d_cols_kernel(float2* __restrict__ z, int s0, int s1, DevPlan plan, const float2* __restrict__ tw,
                                                         const float* __restrict__ mcol, const float* __restrict__ krow_pos)
{
    extern __shared__ float2 lds2d[];
    const int zs = line_stride(plan.n);
    const int x0 = blockIdx.x * G;
    for (int idx = threadIdx.x; idx < s0 * G; idx += blockDim.x) {
        const int y = idx / G, c = idx - y * G, x = x0 + c;
        float2 v = make_float2(0.f, 0.f);
        if (x < s1) {
            v = z[static_cast<size_t>(y) * s1 + x];
            if (CONV) v = cscale(v, krow_pos[x]);
        }
        lds2d[c * zs + phys(y)] = v;
    }
    __syncthreads();
    if (CONV) fftconv_lines<G>(lds2d, zs, plan, tw, mcol);
    else fft_forward_lines<G>(lds2d, zs, plan, tw);
    for (int idx = threadIdx.x; idx < s0 * G; idx += blockDim.x) {
        const int y = idx / G, c = idx - y * G, x = x0 + c;
        if (x < s1) z[static_cast<size_t>(y) * s1 + x] = lds2d[c * zs + phys(y)];
    }
}

__global__ __launch_bounds__(256) void img2d_rows_inv_kernel(const float2* __restrict__ z, uint8_t* __restrict__ dst, float* __restrict__ planes, Img2dGeom g,
                                                             int ca, int cb, DevPlan plan, const float2* __restrict__ tw)
{
    extern __shared__ float2 lds2d[];
    const int zs = line_stride(plan.n);
    const int i = blockIdx.x;                                   // row of the cropped image
    const float2* in = z + static_cast<size_t>(i + g.top) * g.s1;
    for (int pos = threadIdx.x; pos < g.s1; pos += blockDim.x) lds2d[phys(pos)] = in[pos];
    __syncthreads();
    fft_inverse_lines<1>(lds2d, zs, plan, tw);
    const size_t plane = static_cast<size_t>(g.rows) * g.cols;
    for (int j = threadIdx.x; j < g.cols; j += blockDim.x) {
        const float2 v = lds2d[phys(j + g.left)];
        const size_t px = static_cast<size_t>(i) * g.cols + j;
        dst[3 * px + ca] = static_cast<uint8_t>(static_cast<int>(v.x + 0.5f));
        if (cb >= 0) dst[3 * px + cb] = static_cast<uint8_t>(static_cast<int>(v.y + 0.5f));
        if (planes) {
            planes[ca * plane + px] = v.x;
            if (cb >= 0) planes[cb * plane + px] = v.y;
        }
    }
}

__global__ __launch_bounds__(256) void img2d_logspec_kernel(const float2* __restrict__ z, uint8_t* __restrict__ dst, float* __restrict__ planes, Img2dGeom g,
                                                            int ca, int cb, const int* __restrict__ pos_of_freq_y, const int* __restrict__ pos_of_freq_x)
{
    const size_t plane = static_cast<size_t>(g.rows) * g.cols;
    for (size_t px = blockIdx.x * 256u + threadIdx.x; px < plane; px += static_cast<size_t>(gridDim.x) * 256u) {
        const int i = static_cast<int>(px / g.cols), j = static_cast<int>(px - static_cast<size_t>(i) * g.cols);
        const int row = i + g.top, col = j + g.left;
        // FFTSHIFT, "odd/even treated as in matlab" (Source.cpp:239-241)
        const int fy = (row + (g.s0 % 2 == 0 ? g.s0 : g.s0 + 1) / 2) % g.s0;
        const int col_ = (col + (g.s1 % 2 == 0 ? g.s1 : g.s1 + 1) / 2) % g.s1;
        // "Reverse reading from end to the beginning after reached (sizes[1] / 2 + 1)" (:243): the reference indexes
        // its HALF spectrum, so the right half shows column s1/2 - col_ % (s1/2) of the SAME row
        const int fx = col_ < g.s1 / 2 + 1 ? col_ : g.s1 / 2 - col_ % (g.s1 / 2);
        const float2 p = z[static_cast<size_t>(pos_of_freq_y[fy]) * g.s1 + pos_of_freq_x[fx]];
        const float2 q = z[static_cast<size_t>(pos_of_freq_y[(g.s0 - fy) % g.s0]) * g.s1 + pos_of_freq_x[(g.s1 - fx) % g.s1]];
        const float va = 20.f * log10f(fabsf(0.5f * (p.x + q.x)) + 0.00001f);
        dst[3 * px + ca] = static_cast<uint8_t>(static_cast<int>(va + 0.5f));
        if (planes) planes[ca * plane + px] = va;
        if (cb >= 0) {
            const float vb = 20.f * log10f(fabsf(0.5f * (p.y + q.y)) + 0.00001f);
            dst[3 * px + cb] = static_cast<uint8_t>(static_cast<int>(vb + 0.5f));
            if (planes) planes[cb * plane + px] = vb;
        }
    }
}

// 16 bytes per lane, grid-stride: the access shape of the engines' wide loads and stores (blur_copy_bandwidth)
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * 256ull) dst[i] = src[i];
}

// Reflect_101<uint8_t, C> (Utils.hpp:212-243) as a piece: out[(rows+top+bottom) x (cols+left+right) x C]
__global__ __launch_bounds__(256) void reflect101_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int rows, int cols, int C,
                                                            int top, int left, int out_rows, int out_cols)
{
    const size_t total = static_cast<size_t>(out_rows) * out_cols;
    for (size_t px = blockIdx.x * 256u + threadIdx.x; px < total; px += static_cast<size_t>(gridDim.x) * 256u) {
        const int y = static_cast<int>(px / out_cols), x = static_cast<int>(px - static_cast<size_t>(y) * out_cols);
        const uint8_t* s = in + (static_cast<size_t>(refl101(y - top, rows)) * cols + refl101(x - left, cols)) * C;
        for (int c = 0; c < C; ++c) out[px * C + c] = s[c];
    }
}

}  // namespace

// ======================================================================================
// host side: context, caches, launches
// ======================================================================================
struct DevicePlan {
    FftPlan host;
    DevPlan dev{};
    float2* d_tw = nullptr;
    const FastEntry* fast = nullptr;   // compile-time specialised kernels for this length, if any
    int key = 0;
    int* d_pos_of_freq = nullptr;      // inverse of freq_of_pos, uploaded on first use (whole-image 2D path)
};

// staging for blur_gaussian_u8c3_host_batch: frame i+1 travels to the device and frame i-1 back to the host
// while frame i is in the kernels (three slots, two copy streams beside the context's kernel stream)
struct HostPipe {
    static constexpr int S = 3;
    hipStream_t h2d = nullptr, d2h = nullptr;
    uint8_t* buf[S] = { nullptr, nullptr, nullptr };
    size_t bytes = 0;
    hipEvent_t in_done[S] = {}, comp_done[S] = {}, out_done[S] = {};
    bool ready = false;
};

struct MxTables {
    void* frags_row = nullptr;   // [2][nkb][64][8] binary16 (host_math.hpp: mx_fragments)
    void* frags_col = nullptr;
    float* taps_row = nullptr;   // 2 pad + 1 floats, centre at pad (the same taps serve both axes)
    float dr = 0.f, dc = 0.f;    // m[0] - m[N/2] of the reference's row / column transform length (the quirk's gain)
};

struct blur_ctx {
    int device = 0;
    HostPipe pipe;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<int, std::unique_ptr<DevicePlan>> plans;   // key: 4*n + role (0 generic, 1 specialised row, 2 specialised column)
    // (plan key, ksize, quirk, sigma bits) -> device multiplier table in position order
    std::map<std::tuple<int, int, int, uint64_t>, float*> spectra;
    std::map<int, float*> last_spectrum;   // n -> most recent table (diagnostic stamp read-back)
    int num_cus = 256;           // hipDeviceProp_t::multiProcessorCount
    int num_xcds = 8;            // hipDeviceAttributeNumberOfXccs
    // wave-resident engine (wr_kernels.hpp): shared 256-point twiddles, pass-0 twiddles per R0, multiplier tables
    float2* d_w256 = nullptr;
    std::map<int, float2*> wr_tw0;
    std::map<std::tuple<int, int, int, int, uint64_t>, float*> wr_spectra;   // (n, n_ref, ksize | -1, quirk, sigma bits | hash)
    struct LinesTable { uint64_t hash; std::vector<float> host; float* dev; };
    std::list<LinesTable> lines_tables;                                     // blur_convolve_lines_c32_dev: caller tables, LRU, bounded
    // matrix-core engine (mx_kernels.hpp): Toeplitz fragments + taps per kernel, integer sums and float terms of the quirk
    std::map<std::tuple<int, int, int, int, int, uint64_t>, MxTables> mx_tables;   // (ksize | -1, pad, nkb, n_row, n_col, sigma bits | hash)
    uint8_t* fx_strips = nullptr;   // fused kernel: the edge chunks' windows with the mirrored pixels in place
    size_t fx_strips_bytes = 0;
    int* fx_sums = nullptr;         // fused kernel, quirk: the pre-pass's partial sums (run_fx_u8c3)
    size_t fx_sums_bytes = 0;
    // tiled wave-resident path: the quirk's sums and term vectors of one frame, a copy of an in-place frame, the taps per kernel
    int* tl_sums = nullptr;
    size_t tl_sums_bytes = 0;
    float* tl_terms = nullptr;
    size_t tl_terms_bytes = 0;
    uint8_t* tl_copy = nullptr;
    size_t tl_copy_bytes = 0;
    std::map<std::pair<uint64_t, int>, float*> tl_taps;      // (sigma bits, kSize) -> 2 pad + 1 taps on the device
    int* mx_sums = nullptr;
    size_t mx_sums_bytes = 0;
    float* mx_terms = nullptr;
    size_t mx_terms_bytes = 0;
    float* work = nullptr;       // float planes of one frame
    size_t work_bytes = 0;
    float* work2 = nullptr;      // second set of planes: very tall images only (column pass without the LDS pixel stage)
    size_t work2_bytes = 0;
    uint8_t* box_tmp = nullptr;
    size_t box_bytes = 0;
    std::string engine_note;      // BLUR_ENGINE_AUTO: why the last call's choice passed over a faster engine ("" if it did not)
    int last_family = -1;         // kernels the last u8c3 blur used: 0 run-time plans, 1 specialised rows-first, 2 wave-resident, 3 whole-image 2D, 4 matrix-core (two kernels), 6 fused matrix-core
    void* host_stage = nullptr;   // device staging of the host-pointer entry points (kept between calls: no allocation per frame)
    size_t host_stage_bytes = 0;
    int timing = 0;              // 0 off, 1 every timed launch, 2 slot 0 only (blur_ctx_timing_enable)
    std::vector<std::tuple<hipEvent_t, hipEvent_t, int, int>> ev_busy;   // start, stop, kernel (0 row / 1 column), frames
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
    double ms[2] = { 0, 0 };
    int launches[2] = { 0, 0 };
    int frames[2] = { 0, 0 };
};

static thread_local std::string g_create_err;

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return BLUR_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

static int fail(blur_ctx* ctx, int code, const char* msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

static int get_plan(blur_ctx* ctx, int n, bool want_fast, bool column_role, DevicePlan** out)
{
    const FastEntry* fe = want_fast ? find_fast_entry(n, column_role) : nullptr;
    const int key = 4 * n + (fe ? (column_role ? 2 : 1) : 0);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) { *out = it->second.get(); return BLUR_OK; }
    auto dp = std::make_unique<DevicePlan>();
    dp->fast = fe;
    dp->key = key;
    const bool ok = fe ? make_plan_radices(n, fe->radix, fe->npass, dp->host) : make_plan(n, dp->host);
    if (!ok) return fail(ctx, BLUR_ERR_UNSUPPORTED, "FFT length is not 2^a 3^b 5^c");
    dp->dev.n = n;
    dp->dev.npass = dp->host.npass;
    for (int i = 0; i < dp->host.npass; ++i) {
        dp->dev.radix[i] = dp->host.radix[i];
        dp->dev.m[i] = dp->host.m[i];
        dp->dev.tw_off[i] = dp->host.tw_off[i];
    }
    const size_t bytes = std::max<size_t>(dp->host.tw.size() * sizeof(float), 16);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dp->d_tw), bytes));
    if (!dp->host.tw.empty())
        HIP_TRY(ctx, hipMemcpy(dp->d_tw, dp->host.tw.data(), dp->host.tw.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = dp.get();
    ctx->plans[key] = std::move(dp);
    return BLUR_OK;
}

static int get_spectrum(blur_ctx* ctx, const DevicePlan& plan, double sigma, int ksize, bool quirk, float** out)
{
    uint64_t bits;
    std::memcpy(&bits, &sigma, sizeof bits);
    const auto key = std::make_tuple(plan.key, ksize, quirk ? 1 : 0, bits);
    auto it = ctx->spectra.find(key);
    if (it != ctx->spectra.end()) { *out = it->second; return BLUR_OK; }
    const int n = plan.dev.n;
    std::vector<float> m(n / 2 + 1), mp(n);
    kernel_multipliers(sigma, ksize, n, m.data());
    permuted_multipliers(plan.host, m.data(), quirk, mp.data());
    float* d = nullptr;
    // the tail behind the table is only written by -DFK_STAMPS diagnostic builds
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * (n + kStampTailFloats)));
    HIP_TRY(ctx, hipMemset(d, 0, sizeof(float) * (n + kStampTailFloats)));
    HIP_TRY(ctx, hipMemcpy(d, mp.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    ctx->spectra[key] = d;
    ctx->last_spectrum[plan.key] = d;
    *out = d;
    return BLUR_OK;
}

// multiplier table of a caller-supplied n-periodic kernel (box/tent mode, user kernels); cached by content
static int get_spectrum_custom(blur_ctx* ctx, const DevicePlan& plan, const std::vector<float>& karr, bool quirk, float** out)
{
    uint64_t h = 1469598103934665603ull;                         // FNV-1a over the kernel bytes
    const unsigned char* bytes = reinterpret_cast<const unsigned char*>(karr.data());
    for (size_t i = 0; i < karr.size() * sizeof(float); ++i) { h ^= bytes[i]; h *= 1099511628211ull; }
    const auto key = std::make_tuple(plan.key, -1, quirk ? 1 : 0, h);
    auto it = ctx->spectra.find(key);
    if (it != ctx->spectra.end()) { *out = it->second; return BLUR_OK; }
    const int n = plan.dev.n;
    std::vector<float> m(n / 2 + 1), mp(n);
    kernel_multipliers_from_array(karr.data(), n, m.data());
    permuted_multipliers(plan.host, m.data(), quirk, mp.data());
    float* d = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * (n + kStampTailFloats)));
    HIP_TRY(ctx, hipMemset(d, 0, sizeof(float) * (n + kStampTailFloats)));
    HIP_TRY(ctx, hipMemcpy(d, mp.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    ctx->spectra[key] = d;
    ctx->last_spectrum[plan.key] = d;
    *out = d;
    return BLUR_OK;
}

// device staging buffer of the host-pointer entry points, grown on demand and kept in the context
static int ensure_host_stage(blur_ctx* ctx, size_t bytes, void** out)
{
    if (ctx->host_stage_bytes < bytes) {
        if (ctx->host_stage) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->host_stage)); ctx->host_stage = nullptr; ctx->host_stage_bytes = 0; }
        HIP_TRY(ctx, hipMalloc(&ctx->host_stage, bytes ? bytes : 1));
        ctx->host_stage_bytes = bytes;
    }
    *out = ctx->host_stage;
    return BLUR_OK;
}

// The float workspace sits between two guard bands of kWorkGuard bytes filled with 0xA5 when it is allocated;
// blur_debug_check_workspace_guards() counts the guard bytes that no longer hold it (redzone tests).
constexpr size_t kWorkGuard = 4096;
static int ensure_work(blur_ctx* ctx, size_t bytes)
{
    if (ctx->work_bytes >= bytes) return BLUR_OK;
    if (ctx->work) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(reinterpret_cast<char*>(ctx->work) - kWorkGuard));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
    }
    char* base = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&base), bytes + 2 * kWorkGuard));
    HIP_TRY(ctx, hipMemset(base, 0xA5, kWorkGuard));
    HIP_TRY(ctx, hipMemset(base + kWorkGuard + bytes, 0xA5, kWorkGuard));
    ctx->work = reinterpret_cast<float*>(base + kWorkGuard);
    ctx->work_bytes = bytes;
    return BLUR_OK;
}

// ---- event timing -----------------------------------------------------------------
static int timing_drain(blur_ctx* ctx)
{
    for (auto& t : ctx->ev_busy) {
        hipEvent_t a = std::get<0>(t), b = std::get<1>(t);
        HIP_TRY(ctx, hipEventSynchronize(b));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, a, b));
        ctx->ms[std::get<2>(t)] += ms;
        ctx->launches[std::get<2>(t)] += 1;
        ctx->frames[std::get<2>(t)] += std::get<3>(t);
        ctx->ev_free.emplace_back(a, b);
    }
    ctx->ev_busy.clear();
    return BLUR_OK;
}

struct TimedLaunch {
    blur_ctx* ctx; int which; int nframes; hipEvent_t a = nullptr, b = nullptr; bool on;
    TimedLaunch(blur_ctx* c, int w, int nf = 1) : ctx(c), which(w), nframes(nf), on(c->timing == 1 || (c->timing == 2 && w == 0))
    {
        if (!on) return;
        if (ctx->ev_busy.size() >= 8192) timing_drain(ctx);
        if (ctx->ev_free.empty()) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
        } else { a = ctx->ev_free.back().first; b = ctx->ev_free.back().second; ctx->ev_free.pop_back(); }
        (void)hipEventRecord(a, ctx->stream);
    }
    ~TimedLaunch()
    {
        if (!on) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->ev_busy.emplace_back(a, b, which, nframes);
    }
};

// ---- launches ---------------------------------------------------------------------
static size_t row_lds_bytes(int n) { return static_cast<size_t>(line_stride(n)) * sizeof(float2); }
static size_t col_lds_bytes(int n, int C, int rows, int out_bytes_per_px)
{
    return static_cast<size_t>(C) * line_stride(n) * sizeof(float2) + (out_bytes_per_px ? static_cast<size_t>(rows) * 2 * C * out_bytes_per_px : 0) + 16;
}

// threads per workgroup of the run-time-planned kernels by the LDS a workgroup takes (BLUR_GENERIC_THREADS overrides: developer
// knob).  Measured on single images of the reference's sweep (4000 x 6000 sigma 77.5 / 7600 x 11400 sigma 106.8; 256, 512, 1024
// threads): column kernel 908 / 606 / 622 and 3517 / 2673 / 2694 us -- long lines leave room for one or two workgroups per CU and
// four waves per CU hide nothing of the strip gather; row kernel 367 / 459 / 437 and 1641 / 1263 / 1210 us -- better only for the
// longest lines.
static int generic_threads(size_t lds, bool column_pass)
{
    static const int forced = [] { const char* e = getenv("BLUR_GENERIC_THREADS"); return e && *e ? atoi(e) : 0; }();
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    if (column_pass) return lds > 40 * 1024 ? 512 : 256;
    return lds > 80 * 1024 ? 1024 : 256;
}

template <typename K> static int set_lds(blur_ctx* ctx, K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
    return BLUR_OK;
}

template <typename InT, int CH>
static int launch_rowpass(blur_ctx* ctx, const InT* src, float* planes, int rows, int cols, int pad,
                          const DevicePlan& plan, const float* mperm, int gshift = 0)
{
    const size_t lds = row_lds_bytes(plan.dev.n);
    DevPlan dv = plan.dev;
    dv.nxcd = ctx->num_xcds;
    if (lds > kLdsLimit) return fail(ctx, BLUR_ERR_UNSUPPORTED, "row FFT length exceeds LDS capacity");
    const int grid = ((rows + 1) / 2) * CH;
    // long lines leave room for one or two workgroups per CU only: four times the threads then (the passes are loops over
    // butterflies with a barrier each, and 4 waves per CU hide nothing)
    const int threads = generic_threads(lds, false);
    TimedLaunch t(ctx, 0);
    if (threads == 1024) {
        if (int rc = set_lds(ctx, rowpass_kernel<InT, CH, 1024>, lds)) return rc;
        hipLaunchKernelGGL((rowpass_kernel<InT, CH, 1024>), dim3(grid), dim3(1024), lds, ctx->stream, src, planes, rows, cols, pad, dv, plan.d_tw, mperm, gshift);
    } else if (threads == 512) {
        if (int rc = set_lds(ctx, rowpass_kernel<InT, CH, 512>, lds)) return rc;
        hipLaunchKernelGGL((rowpass_kernel<InT, CH, 512>), dim3(grid), dim3(512), lds, ctx->stream, src, planes, rows, cols, pad, dv, plan.d_tw, mperm, gshift);
    } else {
        if (int rc = set_lds(ctx, rowpass_kernel<InT, CH>, lds)) return rc;
        hipLaunchKernelGGL((rowpass_kernel<InT, CH>), dim3(grid), dim3(kThreads), lds, ctx->stream, src, planes, rows, cols, pad, dv, plan.d_tw, mperm, gshift);
    }
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

template <typename OutT, int CH, int C>
static int launch_colpass_c(blur_ctx* ctx, const float* planes, OutT* dst, int rows, int cols, int pad,
                            const DevicePlan& plan, const float* mperm, int strips)
{
    const size_t lds = col_lds_bytes(plan.dev.n, C, rows, sizeof(OutT) == 1 ? CH : 0);
    DevPlan dv = plan.dev;
    dv.nxcd = ctx->num_xcds;
    const int grid = (cols + 2 * C - 1) / (2 * C);
    const int threads = generic_threads(lds, true);
    TimedLaunch t(ctx, 1);
    if (threads == 1024) {
        if (int rc = set_lds(ctx, colpass_kernel<OutT, CH, C, 1024>, lds)) return rc;
        hipLaunchKernelGGL((colpass_kernel<OutT, CH, C, 1024>), dim3(grid), dim3(1024), lds, ctx->stream, planes, dst, rows, cols, pad, dv, plan.d_tw, mperm, strips);
    } else if (threads == 512) {
        if (int rc = set_lds(ctx, colpass_kernel<OutT, CH, C, 512>, lds)) return rc;
        hipLaunchKernelGGL((colpass_kernel<OutT, CH, C, 512>), dim3(grid), dim3(512), lds, ctx->stream, planes, dst, rows, cols, pad, dv, plan.d_tw, mperm, strips);
    } else {
        if (int rc = set_lds(ctx, colpass_kernel<OutT, CH, C>, lds)) return rc;
        hipLaunchKernelGGL((colpass_kernel<OutT, CH, C>), dim3(grid), dim3(kThreads), lds, ctx->stream, planes, dst, rows, cols, pad, dv, plan.d_tw, mperm, strips);
    }
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

// complex lines (pairs of columns) a column-pass workgroup takes: as many as the options ask for and LDS holds; 0 = not even one
static int colpass_lines(int n, int rows, int col_group, int obpp)
{
    int C = col_group > 0 ? col_group / 2 : 4;
    if (C >= 8) C = 8; else if (C >= 4) C = 4; else if (C >= 2) C = 2; else C = 1;
    while (C > 1 && col_lds_bytes(n, C, rows, obpp) > kLdsLimit) C /= 2;
    return col_lds_bytes(n, C, rows, obpp) > kLdsLimit ? 0 : C;
}

// strips: the intermediate is strip-major with strips of 2 C columns (rowpass_kernel's gshift = log2(2 C))
template <typename OutT, int CH>
static int launch_colpass(blur_ctx* ctx, const float* planes, OutT* dst, int rows, int cols, int pad,
                          const DevicePlan& plan, const float* mperm, int col_group, int strips = 0)
{
    const int C = colpass_lines(plan.dev.n, rows, col_group, sizeof(OutT) == 1 ? CH : 0);
    if (!C) return fail(ctx, BLUR_ERR_UNSUPPORTED, "column FFT length exceeds LDS capacity");
    switch (C) {
    case 8: return launch_colpass_c<OutT, CH, 8>(ctx, planes, dst, rows, cols, pad, plan, mperm, strips);
    case 4: return launch_colpass_c<OutT, CH, 4>(ctx, planes, dst, rows, cols, pad, plan, mperm, strips);
    case 2: return launch_colpass_c<OutT, CH, 2>(ctx, planes, dst, rows, cols, pad, plan, mperm, strips);
    default: return launch_colpass_c<OutT, CH, 1>(ctx, planes, dst, rows, cols, pad, plan, mperm, strips);
    }
}

struct Prepared {
    Sizing sz;
    DevicePlan *row = nullptr, *col = nullptr;
    float *m_row = nullptr, *m_col = nullptr;
    int col_group = 0;
    int col_fast_c = 0;     // > 0: complex lines per workgroup of the specialised column kernel
    int tile_w = 0;         // > 0: both passes specialised, the float intermediate uses the strip layout
    size_t frame_elems = 0; // floats of intermediate per frame
    int gen_gshift = 0;     // run-time-planned kernel pair: log2 of the strip width of their strip-major intermediate (0: row-major planes)
    // wave-resident kernels (columns first, then rows) when both passes have one
    const WrEntry *wr_col = nullptr, *wr_row = nullptr;
    float2 *wr_tw0_col = nullptr, *wr_tw0_row = nullptr;
    float *wr_m_col = nullptr, *wr_m_row = nullptr;
    // matrix-core kernels (both passes)
    const MxEntry* mx = nullptr;
    const FxEntry* fx = nullptr;      // fused kernel (fx_kernels.hpp); taps and quirk gains shared with the two-kernel matrix engine
    const MxTables* mxt = nullptr;
    int mx_vpitch = 0;
    bool mx_quirk = false;
    // tiled wave-resident path (run_wr_tiled): bands of rows for the column pass, tiles of columns for the row pass
    struct Span { int in0, in1, padmode, v0, v1; };      // lines [in0, in1) go in (padmode = pad: reflect-101 at both ends; 0: none), [v0, v1) are kept
    std::vector<Span> bands, tiles;
    const WrEntry *tl_col = nullptr;                      // one column kernel for every band
    std::vector<const WrEntry*> tl_row;                   // a row kernel per tile
    std::vector<float*> tl_m_row;
    std::vector<float2*> tl_tw0_row;
    float* tl_m_col = nullptr;
    float2* tl_tw0_col = nullptr;
    float tl_dr = 0.f, tl_dc = 0.f;
    float* tl_taps = nullptr;                             // device: the 2 pad + 1 taps
    bool tiled = false, tl_quirk = false;
};

// a caller-supplied separable kernel instead of the Gaussian: taps (odd count, centre in the middle)
// or, for the tent mode, the reference's box_kernel arrays; pad as the caller's mode defines it
struct CustomKernel {
    const float* taps = nullptr;
    int ksize = 0;
    int pad = 0;
    int box_klen = 0;      // > 0: build the arrays with box_kernel_1d(klen) instead of from taps
};

// ---- wave-resident engine: tables --------------------------------------------------------------
static int wr_get_tables(blur_ctx* ctx, const WrEntry* e, float2** tw0)
{
    if (!ctx->d_w256) {
        std::vector<float> w(512);
        wr_w256(w.data());
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_w256), w.size() * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_w256, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    auto it = ctx->wr_tw0.find(e->r0);
    if (it == ctx->wr_tw0.end()) {
        std::vector<float> t(static_cast<size_t>(e->r0 - 1) * kWrS * 2 + 2);
        wr_tw0(e->r0, t.data());
        float2* d = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), t.size() * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        it = ctx->wr_tw0.emplace(e->r0, d).first;
    }
    *tw0 = it->second;
    return BLUR_OK;
}

// multiplier table (natural order, n floats) of the n-periodic kernel `karr`; cached by `key`
static int wr_get_spectrum(blur_ctx* ctx, const std::tuple<int, int, int, int, uint64_t>& key, const std::vector<float>& karr, int n, int n_ref, bool quirk, float** out)
{
    auto it = ctx->wr_spectra.find(key);
    if (it != ctx->wr_spectra.end()) { *out = it->second; return BLUR_OK; }
    std::vector<float> m(n);
    wr_multipliers(karr.data(), n, n_ref, quirk, m.data());
    float* d = nullptr;
    // the tail behind the table is only written by -DWR_STAMPS diagnostic builds
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * (n + kWrStampTailFloats)));
    HIP_TRY(ctx, hipMemset(d, 0, sizeof(float) * (n + kWrStampTailFloats)));
    HIP_TRY(ctx, hipMemcpy(d, m.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    ctx->wr_spectra[key] = d;
    ctx->last_spectrum[-n] = d;
    *out = d;
    return BLUR_OK;
}

static uint64_t fnv1a(const void* data, size_t bytes)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// ---- matrix-core engine: tables ----------------------------------------------------------------------------------
// taps of one axis from the reference's n-periodic kernel array (centre at index 0): taps[t + pad] = karr[(t + n) % n]
// quiet: a kernel the engine cannot hold is reported by the return value only (the caller falls back to another engine)
static int mx_get_tables(blur_ctx* ctx, int nkb, double sigma, const Sizing& sz, const CustomKernel* ck, const MxTables** out, bool quiet = false)
{
    auto refuse = [&](int code, const char* msg) { return quiet ? code : fail(ctx, code, msg); };
    const int pad = sz.pad;
    std::vector<float> karr[2];                                  // row axis (n_row), column axis (n_col)
    const int nn[2] = { sz.n_row, sz.n_col };
    uint64_t tag;
    int kkey;
    auto build_arrays = [&]() -> int {
        for (int ax = 0; ax < 2; ++ax) {
            const int n = nn[ax];
            if (ck) {
                karr[ax].assign(n, 0.f);
                if (ck->box_klen > 0) box_kernel_1d(karr[ax].data(), ck->box_klen, n);
                else {
                    if (ck->ksize > n) return refuse(BLUR_ERR_INVALID, "kernel longer than the padded line");
                    const int c = ck->ksize / 2;
                    for (int t = 0; t < ck->ksize; ++t) karr[ax][(t - c + n) % n] += ck->taps[t];
                }
            } else {
                karr[ax].assign(std::max(n, sz.kSize), 0.f);
                get_gaussian(karr[ax].data(), sigma, sz.kSize, n);    // Source.cpp:75-102
            }
        }
        return BLUR_OK;
    };
    // the Gaussian's key needs no kernel array: look it up before building one
    if (ck) {
        if (int rc = build_arrays()) return rc;
        kkey = -1;
        tag = fnv1a(karr[0].data(), nn[0] * sizeof(float)) ^ (fnv1a(karr[1].data(), nn[1] * sizeof(float)) * 3);
    } else { kkey = sz.kSize; std::memcpy(&tag, &sigma, sizeof tag); }
    // (pad and nkb are part of the key: the same taps with another pad, or the two-kernel and the fused engine with different
    // window sizes for one pad, have different fragment tables)
    const auto key = std::make_tuple(kkey, pad, nkb, nn[0], nn[1], tag);
    auto it = ctx->mx_tables.find(key);
    if (it != ctx->mx_tables.end()) { *out = &it->second; return BLUR_OK; }
    if (!ck) { if (int rc = build_arrays()) return rc; }
    MxTables t;
    auto release = [&]() {                                        // a failure on the second axis must not leak the first one's tables
        if (t.frags_row) (void)hipFree(t.frags_row);
        if (t.frags_col) (void)hipFree(t.frags_col);
        if (t.taps_row) (void)hipFree(t.taps_row);
    };
    for (int ax = 0; ax < 2; ++ax) {
        const int n = nn[ax];
        std::vector<float> taps(2 * pad + 1);
        for (int k = -pad; k <= pad; ++k) taps[k + pad] = karr[ax][(k + n) % n];
        // anything of the kernel array outside +-pad would be lost here: the Toeplitz band is 2 pad + 1 wide
        for (int i = pad + 1; i < n - pad; ++i)
            if (karr[ax][i] != 0.f) { release(); return refuse(BLUR_ERR_UNSUPPORTED, "matrix-core engine: kernel wider than 2 pad + 1"); }
        // the 24-bit intermediate of the two-kernel engine (mx_kernels.hpp) covers [0, 256): non-negative taps with sum <= 1 keep the
        // row pass inside 0..255.13 (the quirk's terms are added in f32 after it is decoded: no bound on them)
        {
            double sum = 0;
            bool neg = false;
            for (float t : taps) { sum += t; neg = neg || t < 0.f; }
            if (neg || sum > 1.0005) { release(); return refuse(BLUR_ERR_UNSUPPORTED, "matrix-core engine: taps must be non-negative with sum <= 1"); }
        }
        std::vector<uint16_t> fr(static_cast<size_t>(2) * nkb * 512);
        mx_fragments(taps.data(), pad, nkb, fr.data());
        // m[0] and m[n/2] as host_math's kernel_multipliers computes them: float(Re DFT) * (1.f / n)
        long double k0 = 0, kalt = 0;
        for (int i = 0; i < n; ++i) { k0 += karr[ax][i]; kalt += (i & 1) ? -static_cast<long double>(karr[ax][i]) : static_cast<long double>(karr[ax][i]); }
        const float scaler = 1.f / n;
        const float m0 = static_cast<float>(k0) * scaler, mh = static_cast<float>(kalt) * scaler;
        void* dfr = nullptr;
        hipError_t e = hipMalloc(&dfr, fr.size() * sizeof(uint16_t));
        if (e == hipSuccess) {
            (ax ? t.frags_col : t.frags_row) = dfr;
            e = hipMemcpy(dfr, fr.data(), fr.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && ax == 0) {                        // the taps themselves: the fused engine's quirk terms convolve with them
            e = hipMalloc(reinterpret_cast<void**>(&t.taps_row), taps.size() * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(t.taps_row, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            release();
            (void)hipGetLastError();          // clear it: under AUTO the caller falls back to another engine, whose first launch check would pick it up
            if (!quiet) ctx->err = std::string("matrix-core tables: ") + hipGetErrorString(e);
            return BLUR_ERR_HIP;
        }
        (ax ? t.dc : t.dr) = m0 - mh;
    }
    *out = &(ctx->mx_tables[key] = t);
    return BLUR_OK;
}

// ptrs_aligned: the frame pointers of the call are 4-byte aligned (no engine depends on it any more: the fused kernels take any
// width and alignment since round 4)
// The one rule both the matrix-core branch and the FFT branch of prepare() consult: do the wave-resident FFT kernels pay for this
// frame?  Both passes need a kernel (N = 256 R0 >= line + 2 pad) whose LDS holds the image; where the rows-first family has
// compile-time kernels for BOTH reference lengths as well (the six BASELINE (length, role) pairs) the wave-resident pair must fill
// its waves (R0 >= 8 / 12, lines at least 3/4 of N), elsewhere it competes with the run-time-planned kernels and wins as soon as the
// lines are at least half of N.  (Measured, us per frame wave-resident / rows-first: 4K sigma 20 106 / 107; 1080p sigma 20 41 / 36;
// 1000 x 1500 sigma 38.7 32 / 144; 1300 x 1950 sigma 44 83 / 87.)
struct FftFamilyChoice {
    const WrEntry* wc = nullptr;       // wave-resident column kernel / row kernel, where they exist and fit
    const WrEntry* wr = nullptr;
    bool fits = false;                 // both exist and their LDS holds the image
    bool old_both = false;             // the rows-first family has compile-time kernels for both reference lengths
    bool wr_pays = false;
};
static FftFamilyChoice fft_family_choice(int rows, int cols, const Sizing& sz)
{
    FftFamilyChoice c;
    c.wc = find_wr_entry(rows + 2 * sz.pad, true);
    c.wr = find_wr_entry(cols + 2 * sz.pad, false);
    c.old_both = find_fast_entry(sz.n_col, true) && find_fast_entry(sz.n_row, false);
    c.fits = c.wc && c.wr && c.wc->col_lds(rows) <= kLdsLimit && c.wr->row_lds(cols) <= kLdsLimit;
    if (c.fits) {
        const int need_c = rows + 2 * sz.pad, need_r = cols + 2 * sz.pad, nc = c.wc->r0 * kWrS, nr = c.wr->r0 * kWrS;
        c.wr_pays = c.old_both ? (c.wc->r0 >= 8 && c.wr->r0 >= 12 && 4 * need_c >= 3 * nc && 4 * need_r >= 3 * nr) : (2 * need_c >= nc && 2 * need_r >= nr);
    }
    return c;
}

// ---- tiled wave-resident path: spans along one axis --------------------------------------------------------------------------
// Cuts `len` lines with a kernel of half width `pad` into spans a transform of at most `nmax` points and `in_max` input lines can take.
// The first and the last span end at an image border and are padded by the kernel itself (reflect-101, padmode = pad: in + 2 pad
// points); the spans between are circular convolutions of real pixels only (padmode = 0: in points) whose first `halo` (= pad rounded
// up to `align`, so that the kept part starts at a multiple of `align`) and last pad results are not kept.  False: pad too wide.
static bool plan_spans(int len, int pad, int nmax, int in_max, int align, std::vector<Prepared::Span>& out, bool equal_middle = false, bool edge_max = false)
{
    out.clear();
    if (in_max > nmax) in_max = nmax;
    if (len + 2 * pad <= nmax && len <= in_max) { out.push_back({ 0, len, pad, 0, len }); return true; }
    const int halo = (pad + align - 1) / align * align;
    const int le = std::min(nmax - 2 * pad, in_max), lm = in_max;
    int ve = (le - halo) / align * align, vm = (lm - halo - pad) / align * align;        // most kept lines of an edge / a middle span
    if (ve <= 0 || vm <= 0) return false;
    // boundaries: as few spans as possible, evened out (one or two more where the even split breaks a limit)
    int nsp0 = 2;
    if (len > 2 * ve) nsp0 += (len - 2 * ve + vm - 1) / vm;
    std::vector<int> b;
    int nsp = 0, sm_equal = 0;
    for (int tries = 0; tries < 4 && nsp == 0; ++tries) {
        const int n = nsp0 + tries;
        b.assign(n + 1, 0);
        b[n] = len;
        // edge spans take the same share as the middle ones where that fits their limit
        int e = edge_max ? ve : static_cast<int>(static_cast<long long>(len) / n) / align * align;      // (edge_max: the edge spans as long as they go)
        if (e > ve) e = ve;
        if (2 * e >= len) e = (len / 2) / align * align;
        if (e < align) e = align;
        b[1] = e;
        const int rest = len - 2 * e;
        if (equal_middle && n > 2) {
            // middle spans of ONE length (a multiple of 2 and of align), so that they can run as the "frames" of one launch; the last of
            // them may reach into the last span's rows (written twice, by both)
            const int unit = align % 2 == 0 ? align : 2 * align;
            const int sm = ((rest + (n - 2) - 1) / (n - 2) + unit - 1) / unit * unit;
            if (sm > vm || e + (n - 2) * sm + pad > len) continue;
            for (int i = 2; i < n; ++i) b[i] = e + (i - 1) * sm;
            b[n - 1] = len - e;                                                               // the last span keeps its own e lines
            if (b[n - 1] <= b[n - 2] - sm) continue;
            nsp = n;
            sm_equal = sm;
            break;
        }
        for (int i = 2; i < n; ++i) b[i] = (e + static_cast<int>(static_cast<long long>(rest) * (i - 1) / (n - 2))) / align * align;
        b[n - 1] = std::max(b[n - 1], (len - ve + align - 1) / align * align);             // the last span's limit
        bool ok = len - b[n - 1] <= ve && len - b[n - 1] > 0 && b[1] <= ve;
        for (int i = 1; i < n && ok; ++i) ok = b[i] > b[i - 1];
        for (int i = 1; i + 1 < n && ok; ++i) ok = b[i + 1] - b[i] <= vm;
        if (ok) nsp = n;
    }
    if (nsp == 0) return false;
    for (int i = 0; i < nsp; ++i) {
        Prepared::Span sp;
        sp.v0 = b[i];
        sp.v1 = b[i + 1];
        if (i == 0) { sp.in0 = 0; sp.in1 = std::min(len, sp.v1 + pad); sp.padmode = pad; }
        else if (i == nsp - 1) { sp.in0 = sp.v0 - halo; sp.in1 = len; sp.padmode = pad; }
        else {
            if (sm_equal > 0) sp.v1 = sp.v0 + sm_equal;            // (the last middle span may reach into the last span's lines)
            sp.in0 = sp.v0 - halo; sp.in1 = std::min(len, sp.v1 + pad); sp.padmode = 0;
        }
        if (sp.in0 < 0 || sp.in1 - sp.in0 + 2 * sp.padmode > nmax || sp.in1 - sp.in0 > in_max || sp.padmode > sp.in1 - sp.in0 - 1) return false;
        out.push_back(sp);
    }
    return true;
}

// m[0] - m[n / 2] of the n-periodic kernel array as host_math's kernel_multipliers computes the two: float(Re DFT) * (1.f / n)
static float quirk_gain(const float* karr, int n)
{
    long double k0 = 0, kalt = 0;
    for (int i = 0; i < n; ++i) { k0 += karr[i]; kalt += (i & 1) ? -static_cast<long double>(karr[i]) : static_cast<long double>(karr[i]); }
    const float scaler = 1.f / n;
    return static_cast<float>(k0) * scaler - static_cast<float>(kalt) * scaler;
}

// the tiled wave-resident path for this frame: the column kernel and its bands, the row kernels and their tiles, the tables
// (multipliers WITHOUT the quirk: it enters as the terms of tl_terms_kernel).  tile_points > 0 (tests): no transform longer than that.
static int plan_tiled(blur_ctx* ctx, int rows, int cols, double sigma, bool quirk, int tile_points, Prepared& p)
{
    const int pad = p.sz.pad;
    // one column kernel for all bands: the candidate that transforms the fewest points per column
    long long best = -1;
    for (int r0 : { 16, 15, 12, 10, 9, 8, 6, 5, 4, 3 }) {
        const int n = r0 * kWrS;
        if (tile_points > 0 && n > tile_points) continue;
        const WrEntry* e = find_wr_entry(n, true);
        if (!e || e->r0 != r0) continue;
        int in_max = n;
        while (in_max > 0 && e->col_lds(in_max) > kLdsLimit) in_max -= 8;
        std::vector<Prepared::Span> sp;
        if (in_max <= 0 || !plan_spans(rows, pad, n, in_max, 1, sp, true)) continue;
        const long long cost = static_cast<long long>(sp.size()) * n;
        if (best < 0 || cost < best) { best = cost; p.bands = sp; p.tl_col = e; }
    }
    if (best < 0) return BLUR_ERR_UNSUPPORTED;
    best = -1;
    for (int r0 : { 16, 15, 12, 10, 9, 8, 6, 5, 4, 3 }) {
        const int n = r0 * kWrS;
        if (tile_points > 0 && n > tile_points) continue;
        const WrEntry* e = find_wr_entry(n, false);
        if (!e || e->r0 != r0) continue;
        int in_max = n;
        while (in_max > 0 && e->row_lds(in_max) > kLdsLimit) in_max -= 16;
        if (in_max <= 0) continue;
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<Prepared::Span> sp;
            if (!plan_spans(cols, pad, n, in_max, 16, sp, false, variant == 1)) continue;
            // every tile runs on the smallest kernel that holds it: the cost is the transform points per row
            long long cost = 0;
            for (const Prepared::Span& t : sp) {
                const WrEntry* te = find_wr_entry(t.in1 - t.in0 + 2 * t.padmode, false);
                cost += te ? te->r0 * kWrS : 1 << 20;
            }
            if (best < 0 || cost < best) { best = cost; p.tiles = sp; }
        }
    }
    if (best < 0) return BLUR_ERR_UNSUPPORTED;
    uint64_t bits;
    std::memcpy(&bits, &sigma, sizeof bits);
    auto spectrum = [&](int n, float** out) -> int {
        const auto key = std::make_tuple(n, 0, p.sz.kSize, 0, bits);            // n_ref 0: no quirk in the table
        std::vector<float> karr(n, 0.f);
        if (ctx->wr_spectra.find(key) == ctx->wr_spectra.end()) {
            std::vector<float> k(std::max(n, p.sz.kSize));
            get_gaussian(k.data(), sigma, p.sz.kSize, n);                       // Source.cpp:75-102: taps rotated to index 0
            std::copy(k.begin(), k.begin() + n, karr.begin());
        }
        return wr_get_spectrum(ctx, key, karr, n, n, false, out);
    };
    if (int rc = wr_get_tables(ctx, p.tl_col, &p.tl_tw0_col)) return rc;
    if (int rc = spectrum(p.tl_col->r0 * kWrS, &p.tl_m_col)) return rc;
    p.tl_row.clear(); p.tl_m_row.clear(); p.tl_tw0_row.clear();
    for (const Prepared::Span& t : p.tiles) {
        const WrEntry* e = find_wr_entry(t.in1 - t.in0 + 2 * t.padmode, false);
        while (e && e->row_lds(t.in1 - t.in0) > kLdsLimit) e = find_wr_entry(e->r0 * kWrS + 1, false);
        if (!e || (tile_points > 0 && e->r0 * kWrS > tile_points)) return BLUR_ERR_UNSUPPORTED;
        float2* tw0 = nullptr;
        float* m = nullptr;
        if (int rc = wr_get_tables(ctx, e, &tw0)) return rc;
        if (int rc = spectrum(e->r0 * kWrS, &m)) return rc;
        p.tl_row.push_back(e); p.tl_tw0_row.push_back(tw0); p.tl_m_row.push_back(m);
    }
    // the quirk's gains at the REFERENCE's transform lengths (Source.cpp:420-425), and the taps
    {
        std::vector<float> k(std::max(std::max(p.sz.n_row, p.sz.n_col), p.sz.kSize));
        get_gaussian(k.data(), sigma, p.sz.kSize, p.sz.n_row);
        p.tl_dr = quirk_gain(k.data(), p.sz.n_row);
        std::vector<float> taps(2 * pad + 1);
        for (int t = -pad; t <= pad; ++t) taps[t + pad] = k[(t + p.sz.n_row) % p.sz.n_row];
        get_gaussian(k.data(), sigma, p.sz.kSize, p.sz.n_col);
        p.tl_dc = quirk_gain(k.data(), p.sz.n_col);
        const auto tk = std::make_pair(bits, p.sz.kSize);
        auto it = ctx->tl_taps.find(tk);
        if (it == ctx->tl_taps.end()) {
            float* d = nullptr;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), taps.size() * sizeof(float)));
            HIP_TRY(ctx, hipMemcpy(d, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice));
            it = ctx->tl_taps.emplace(tk, d).first;
        }
        p.tl_taps = it->second;
    }
    p.frame_elems = 0;
    for (const Prepared::Span& b : p.bands) p.frame_elems = std::max(p.frame_elems, wr_frame_floats(b.in1 - b.in0, cols, b.padmode, p.tl_col->g));
    p.tiled = true;
    p.tl_quirk = quirk;
    return BLUR_OK;
}

static int prepare(blur_ctx* ctx, int rows, int cols, double sigma, const blur_opts* opts, Prepared& p, bool u8c3 = true,
                   const CustomKernel* ck = nullptr, bool allow_wr = true, bool ptrs_aligned = false)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (rows <= 0 || cols <= 0 || (!ck && !(sigma > 0))) return fail(ctx, BLUR_ERR_INVALID, "rows, cols and sigma must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->engine_note.clear();
    if (ck) {
        p.sz = Sizing{};
        p.sz.kSize = ck->box_klen > 0 ? ck->box_klen : ck->ksize;
        p.sz.pad = ck->pad;
        auto fit = [](int want, int& tz) { const int n = is_valid_size(want) ? want : nearest_transform_size(want); tz = n - want; return n; };
        p.sz.n_col = fit(rows + 2 * ck->pad, p.sz.tz_col);          // Source.cpp:445-457
        p.sz.n_row = fit(cols + 2 * ck->pad, p.sz.tz_row);
    } else
    p.sz = pffft_sizing(rows, cols, sigma);
    if (p.sz.pad > rows - 1 || p.sz.pad > cols - 1)
        return fail(ctx, BLUR_ERR_UNSUPPORTED, "pad > min(rows, cols) - 1: reflect-101 would read outside the image (README.md:33-38)");
    const bool quirk = opts ? opts->nyquist_quirk != 0 : true;
    p.col_group = opts ? opts->col_group : 0;
    const bool allow_fast = u8c3 && !(opts && opts->force_generic == 1);   // force the generic kernels (tests)
    // Engine choice (enum blur_engine).  The library's own policy, BLUR_ENGINE_AUTO, in the order it is applied:
    //   | engine                      | where                                                   | measured (MI355X, 8 frames per call)            |
    //   | fused matrix-core kernel    | pad <= 72 (sigma <~ 22), non-negative taps with sum <= 1 | 4K sigma 20: 45 us per frame against 70 for the  |
    //   |                             | (any width, any pointer alignment)                      | two kernels and 105 for the FFT kernels          |
    //   | fused kernel, wide windows  | pad 73 .. 168 on frames >= 1 MP (not where the next row | 4K sigma 30: 98 GP/s against 85 / 70; 8K sigma 40 |
    //   |                             | applies), same conditions                               | single frame 79 against 54 / 26                   |
    //   | FFT, compile-time families  | frames < 1 MP, or pad > 136 on frames >= 6 MP           | 1080p sigma 20 single frame 72 / 78 us;          |
    //   |                             |                                                         | 4K sigma 50: 72 GP/s against 63                  |
    //   | two-kernel matrix engine    | pad <= 168, non-negative taps with sum <= 1             | 4K sigma 26 / 30 / 36: 103 / 87 / 80 GP/s (FFT 76)|
    //   | FFT kernels                 | everything else                                         |                                                   |
    // The choice depends on the frame and the kernel only, never on the number of frames.
    const int choice = opts ? opts->engine : BLUR_ENGINE_AUTO;
    // (allow_wr = false: the caller wants the rows-first float planes themselves, blur_rowpass_u8c3_dev)
    // Where the library's own choice (reserved[3] = 0) stays with the FFT kernels although a matrix-core kernel exists -- only
    // when the FFT engine has a compile-time family that its own policy below would pick (the run-time-planned kernels are
    // 3-4 times slower: then the matrix cores win regardless):
    //  * small frames (< 1 MP): the matrix-core path is four launches against two, and one small frame does not fill the chip
    //    either way (one frame per call, us: 1080p sigma 20 78 matrix / 72 FFT; eight per call: 23 / 36 per frame);
    //  * wide kernels (19 window blocks and more, pad > 136): the row kernel's window is then 3-4 times its 128 output pixels and
    //    its fragments leave one workgroup per CU (4K, 8 frames per call, GP/s matrix / FFT: sigma 36 80 / 76, sigma 40 72 / 73,
    //    sigma 44 66 / 72, sigma 50 63 / 72).
    // The choice depends on the frame and the kernel only, never on the number of frames: a frame blurred alone and the same
    // frame inside a batch give the same bytes.
    bool small_fft = false;
    if (choice == 0) {
        const MxEntry* me0 = find_mx_entry(p.sz.pad);
        // (wide: measured on 4K frames only; the 1.5-3.8 MP images of the reference's sweep with sigma 39-49 run 1.0-1.5 times
        // FASTER on the matrix cores than on the wave-resident FFT kernels, so the rule is limited to large frames)
        const bool small = static_cast<long long>(rows) * cols < 1000000ll, wide = me0 && me0->nkb >= 19 && static_cast<long long>(rows) * cols >= 6000000ll;
        if (small || wide) {
            const FftFamilyChoice fc = fft_family_choice(rows, cols, p.sz);
            small_fft = fc.wr_pays || fc.old_both;
        }
    }
    const bool force_tiled = opts && opts->tile_points > 0;       // (tests: straight to the tiled wave-resident path)
    if (!force_tiled && allow_fast && allow_wr && (choice == BLUR_ENGINE_FUSED || choice == BLUR_ENGINE_AUTO)) {
        const FxEntry* fe = find_fx_entry(p.sz.pad);
        // the one-channel-per-workgroup kernels for wide windows (fw_kernels.hpp; measured against the two-kernel engine / the FFT
        // kernels, GP/s: 4K sigma 30 single frame 61 / 50 / 57, eight frames 98 / 85 / 70; 8K sigma 40 single 79 / 54 / 26;
        // 1080p sigma 30 single 29 / 31 / 36, eight 68 / 63 / 59; 1000 x 1500 sigma 38.7 single 17 / 21 / 13; 4K sigma 50 single
        // 39 / 37 / 57, eight 68-73 / 63 / 63-74, sigma 44 eight 76 / 63 / 64): the library's own choice for frames of 6 MP and more,
        // except the widest window (23 blocks, pad > 152) where the FFT engine has a compile-time family for the frame (small_fft
        // above).
        const char* why = nullptr;
        if (fe && fe->nkb > 11 && choice == BLUR_ENGINE_AUTO) {
            // (round 4, segments of any number of tiles: one image per call, GP/s fused / two kernels / FFT: 1000 x 1500 sigma 38.7 21 / 21 / 16,
            // 1600 x 2400 sigma 49 33 / 27 / 32, 1080p sigma 30 35 / 31 / 37, eight of them 69 / 63 / 63: from 1 MP on; rounds 3-4 had 6 MP here)
            if (static_cast<long long>(rows) * cols < 1000000ll) { fe = nullptr; why = "wide fused kernel (pad 73 .. 168): frames below 1 MP run on two kernels or the FFT kernels"; }
            else if (small_fft && fe->nkb >= 23) { fe = nullptr; why = "wide fused kernel: pad > 152 where the FFT engine has a compile-time family for the frame"; }
        }
        if (!fe && !why) why = "fused matrix-core engine: no kernel instantiated for this pad";
        // (0xfffffff0 is the offset the kernels give a dropped store: it must lie outside the frame's buffer resource)
        // (and the left chunk's image resource starts 3 pad bytes before the frame: its size is the frame's plus those)
        else if (static_cast<long long>(rows) * cols * 3 > 0xfffff000ll) why = "fused matrix-core engine: frame too large for 32-bit offsets";
        else if (quirk && fx_groups_per_thread(cols) == 0) why = "fused matrix-core engine: image wider than 16384 pixels (the quirk's pre-pass)";
        else if (quirk && cols < 4) why = "fused matrix-core engine: image narrower than 4 pixels (the quirk's pre-pass)";
        if (why && choice == BLUR_ENGINE_FUSED) return fail(ctx, BLUR_ERR_UNSUPPORTED, why);
        if (why) ctx->engine_note = why;
        if (!why) {
            const int rc = mx_get_tables(ctx, fe->nkb, sigma, p.sz, ck, &p.mxt, choice == BLUR_ENGINE_AUTO);
            if (rc != BLUR_OK && choice == BLUR_ENGINE_FUSED) return rc;
            if (rc == BLUR_OK) {
                p.fx = fe;
                p.mx_quirk = quirk;
                p.frame_elems = 0;
                return BLUR_OK;
            }
        }
    }
    if (!force_tiled && allow_fast && allow_wr && !small_fft && (choice == 0 || choice == 3)) {
        const MxEntry* me = find_mx_entry(p.sz.pad);
        const int vpitch = (3 * cols + 31) & ~31;
        const bool fits = me && static_cast<long long>(mx_vrows(rows, me->nkb)) * vpitch < (1ll << 30) && static_cast<long long>(rows) * cols * 3 < (1ll << 32);
        if (!fits && choice != 0)
            return fail(ctx, BLUR_ERR_UNSUPPORTED, me ? "matrix-core engine: frame too large for 32-bit element offsets" : "matrix-core engine: no kernel instantiated for this pad");
        int rc_tables = BLUR_OK;
        if (fits) rc_tables = mx_get_tables(ctx, me->nkb, sigma, p.sz, ck, &p.mxt, choice == BLUR_ENGINE_AUTO);
        if (fits && rc_tables != BLUR_OK && choice != 0) return rc_tables;
        if (fits && rc_tables == BLUR_OK) {
        p.mx = me;
        p.mx_vpitch = vpitch;
        p.mx_quirk = quirk;
        p.frame_elems = static_cast<size_t>(mx_vrows(rows, me->nkb)) * vpitch;
        return BLUR_OK;
        }
    }
    // Tiled wave-resident path (round 4): lines longer than the longest wave-resident transform -- sigma = sqrt(side) on large images,
    // the reference's own benchmark (Source.cpp:627-635) -- in bands and tiles through the same kernels, instead of the run-time-planned
    // kernels (three to four times slower).  Taken where neither the whole-image wave-resident pair nor the rows-first family's
    // compile-time kernels apply; blur_opts.tile_points > 0 forces it (tests).
    {
        const int tile_points = opts ? opts->tile_points : 0;
        if (allow_fast && allow_wr && u8c3 && !ck && !(opts && opts->engine == BLUR_ENGINE_FFT_ROWS_FIRST)) {
            const FftFamilyChoice fc0 = fft_family_choice(rows, cols, p.sz);
            const bool whole = fc0.fits && ((opts && opts->engine == BLUR_ENGINE_FFT_WAVE_RESIDENT) || fc0.wr_pays);
            // (measured, one image per call, tools/tiled_compare.py: from 17 MP on the tiled path is 1.06 .. 1.76 times faster than the
            // run-time-planned kernels in either orientation; between 10 and 16 MP only where the columns still fit one transform
            // (rows + 2 pad <= 4096: wide images, 29 .. 35 GP/s against 26 .. 28) -- the reference's own sweep has tall images, whose
            // columns then need two bands of 4096 points each: 5 .. 9 % slower than the run-time-planned kernels there)
            const long long mp = static_cast<long long>(rows) * cols;
            const bool pays = mp >= 16000000ll || (mp >= 10000000ll && rows + 2 * p.sz.pad <= 4096);
            if (tile_points > 0 || (!whole && !fc0.old_both && pays)) {
                const int rc = plan_tiled(ctx, rows, cols, sigma, quirk, tile_points, p);
                if (rc == BLUR_OK) return BLUR_OK;
                if (rc != BLUR_ERR_UNSUPPORTED) return rc;
                if (tile_points > 0) return fail(ctx, BLUR_ERR_UNSUPPORTED, "tiled wave-resident path: the kernel is too wide for transforms of tile_points points");
                p.bands.clear(); p.tiles.clear();
            }
        }
    }
    // Wave-resident kernels first (reserved[3] = 1 switches them off): both passes need one, and its LDS must hold the image
    if (allow_fast && allow_wr && !(opts && opts->engine == BLUR_ENGINE_FFT_ROWS_FIRST)) {
        // (BLUR_ENGINE_FFT_WAVE_RESIDENT: wherever the image fits; otherwise the rule of fft_family_choice)
        const FftFamilyChoice fc = fft_family_choice(rows, cols, p.sz);
        const WrEntry* wc = fc.wc;
        const WrEntry* wr = fc.wr;
        const bool pays = (opts && opts->engine == BLUR_ENGINE_FFT_WAVE_RESIDENT) || fc.wr_pays;
        if (fc.fits && pays &&
            wr_frame_floats(rows, cols, p.sz.pad, wc->g) / 3 < (static_cast<size_t>(1) << 30)) {
            if (int rc = wr_get_tables(ctx, wc, &p.wr_tw0_col)) return rc;
            if (int rc = wr_get_tables(ctx, wr, &p.wr_tw0_row)) return rc;
            for (int pass = 0; pass < 2; ++pass) {
                const WrEntry* e = pass ? wc : wr;
                const int n = e->r0 * kWrS, n_ref = pass ? p.sz.n_col : p.sz.n_row;
                std::vector<float> karr(n, 0.f);
                std::tuple<int, int, int, int, uint64_t> key;
                if (ck) {
                    if (ck->box_klen > 0) box_kernel_1d(karr.data(), ck->box_klen, n);
                    else {
                        if (ck->ksize > n) return fail(ctx, BLUR_ERR_INVALID, "kernel longer than the padded line");
                        const int c = ck->ksize / 2;
                        for (int t = 0; t < ck->ksize; ++t) karr[(t - c + n) % n] += ck->taps[t];
                    }
                    key = std::make_tuple(n, n_ref, -1, quirk ? 1 : 0, fnv1a(karr.data(), karr.size() * sizeof(float)));
                } else {
                    uint64_t bits;
                    std::memcpy(&bits, &sigma, sizeof bits);
                    key = std::make_tuple(n, n_ref, p.sz.kSize, quirk ? 1 : 0, bits);
                    if (ctx->wr_spectra.find(key) == ctx->wr_spectra.end()) {
                        std::vector<float> k(std::max(n, p.sz.kSize));
                        get_gaussian(k.data(), sigma, p.sz.kSize, n);          // Source.cpp:75-102: taps rotated to index 0
                        std::copy(k.begin(), k.begin() + n, karr.begin());
                    }
                }
                if (int rc = wr_get_spectrum(ctx, key, karr, n, n_ref, quirk, pass ? &p.wr_m_col : &p.wr_m_row)) return rc;
            }
            p.wr_col = wc;
            p.wr_row = wr;
            p.frame_elems = wr_frame_floats(rows, cols, p.sz.pad, wc->g);
            return BLUR_OK;
        }
    }
    // the specialised column kernel needs its strip (C complex lines + pixel stage) to fit in LDS
    if (allow_fast)
        if (const FastEntry* fe = find_fast_entry(p.sz.n_col, true)) {
            int C = p.col_group > 0 ? (p.col_group >= 8 ? 4 : 2) : 4;
            while (C >= 2 && fe->col_lds_bytes(rows, C) > kLdsLimit) C /= 2;
            p.col_fast_c = C >= 2 ? C : 0;
        }
    // the specialised row kernel addresses a frame's three float planes with 32-bit byte offsets (fk_launch_row_u8)
    const bool row_fast_ok = static_cast<size_t>(rows + 1) * (cols + 8) * 12 < (static_cast<size_t>(1) << 32) && rows < (1 << 20);
    if (int rc = get_plan(ctx, p.sz.n_row, allow_fast && row_fast_ok, false, &p.row)) return rc;
    if (int rc = get_plan(ctx, p.sz.n_col, allow_fast && p.col_fast_c > 0, true, &p.col)) return rc;
    if (ck) {
        for (int pass = 0; pass < 2; ++pass) {
            DevicePlan* pl = pass ? p.col : p.row;
            const int n = pl->dev.n;
            std::vector<float> karr(n, 0.f);
            if (ck->box_klen > 0) box_kernel_1d(karr.data(), ck->box_klen, n);
            else {
                if (ck->ksize > n) return fail(ctx, BLUR_ERR_INVALID, "kernel longer than the padded line");
                const int c = ck->ksize / 2;
                for (int t = 0; t < ck->ksize; ++t) karr[(t - c + n) % n] += ck->taps[t];   // centre at index 0 (README.md:93-101)
            }
            if (int rc = get_spectrum_custom(ctx, *pl, karr, quirk, pass ? &p.m_col : &p.m_row)) return rc;
        }
    } else {
        if (int rc = get_spectrum(ctx, *p.row, sigma, p.sz.kSize, quirk, &p.m_row)) return rc;
        if (int rc = get_spectrum(ctx, *p.col, sigma, p.sz.kSize, quirk, &p.m_col)) return rc;
    }
    // strip layout of the intermediate only when both kernels understand it
    const bool no_tile = opts && opts->row_major_planes == 1;
    p.tile_w = (p.row->fast && p.col->fast && p.col_fast_c == 4 && !no_tile) ? 8 : 0;
    // strip layout: [strip][row pair][8 columns][2 rows] per channel
    p.frame_elems = p.tile_w ? static_cast<size_t>((cols + p.tile_w - 1) / p.tile_w) * ((rows + 1) / 2) * (2 * p.tile_w) * 3
                             : static_cast<size_t>(rows) * cols * 3;
    // both kernels run-time-planned: strip-major intermediate, strips as wide as a column-pass workgroup's share (the column kernel
    // then reads one contiguous block instead of 16 bytes out of every row: 4000 x 6000 sigma 77.5: 593 -> 4xx us)
    p.gen_gshift = 0;
    if (u8c3 && !p.row->fast && !p.col->fast && !no_tile) {
        int C = colpass_lines(p.col->dev.n, rows, p.col_group, 3);
        if (C < 1) C = colpass_lines(p.col->dev.n, rows, p.col_group, 0);        // very tall images: the column pass runs plane by plane (run_colpass_u8c3)
        if (C >= 1) {
            int g = 1;
            while ((1 << g) < 2 * C) ++g;
            p.gen_gshift = g;
            const int G = 1 << g;
            p.frame_elems = static_cast<size_t>((cols + G - 1) / G) * G * rows * 3;
        }
    }
    return BLUR_OK;
}

// nframes frames back to back (planes: 3*rows*cols floats per frame)
static int run_rowpass_u8c3(blur_ctx* ctx, const uint8_t* src, float* planes, int rows, int cols, int nframes, const Prepared& p, int tile_w)
{
    if (p.row->fast) {
        TimedLaunch t(ctx, 0, nframes);
        HIP_TRY(ctx, p.row->fast->row_u8(ctx->stream, src, planes, rows, cols, p.sz.pad, nframes, tile_w < 0 ? 0 : tile_w, p.row->d_tw, p.m_row));
        return BLUR_OK;
    }
    const size_t px = static_cast<size_t>(rows) * cols;
    const int gshift = tile_w < 0 ? 0 : p.gen_gshift;                 // (tile_w < 0: the caller wants row-major planes)
    const size_t fstride = gshift ? p.frame_elems : px * 3;
    for (int f = 0; f < nframes; ++f)
        if (int rc = launch_rowpass<uint8_t, 3>(ctx, src + f * px * 3, planes + f * fstride, rows, cols, p.sz.pad, *p.row, p.m_row, gshift)) return rc;
    return BLUR_OK;
}

static int run_colpass_u8c3(blur_ctx* ctx, const float* planes, uint8_t* dst, int rows, int cols, int nframes, const Prepared& p)
{
    if (p.col->fast) {
        TimedLaunch t(ctx, 1, nframes);
        HIP_TRY(ctx, p.col->fast->col_u8(ctx->stream, planes, dst, rows, cols, p.sz.pad, nframes, p.tile_w ? 1 : 0, p.col->d_tw, p.m_col, p.col_fast_c));
        return BLUR_OK;
    }
    const size_t px = static_cast<size_t>(rows) * cols;
    if (col_lds_bytes(p.col->dev.n, 1, rows, 3) > kLdsLimit && col_lds_bytes(p.col->dev.n, 1, rows, 0) <= kLdsLimit) {
        // Very tall images: one complex line plus the u8 pixel stage no longer fit in LDS.  Run the
        // column pass plane by plane into float planes (no stage) and interleave afterwards:
        // 24 B/px more traffic, but the FFT length limit becomes that of a bare line (N <~ 19800).
        if (ctx->work2_bytes < px * 3 * sizeof(float)) {
            if (ctx->work2) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->work2)); ctx->work2 = nullptr; ctx->work2_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->work2), px * 3 * sizeof(float)));
            ctx->work2_bytes = px * 3 * sizeof(float);
        }
        const size_t fstride = p.gen_gshift ? p.frame_elems : px * 3, cstride = fstride / 3;
        for (int f = 0; f < nframes; ++f) {
            for (int c = 0; c < 3; ++c)
                if (int rc = launch_colpass<float, 1>(ctx, planes + f * fstride + c * cstride, ctx->work2 + c * px, rows, cols, p.sz.pad, *p.col, p.m_col, p.col_group,
                                                      p.gen_gshift ? 1 : 0)) return rc;
            const unsigned grid = static_cast<unsigned>(std::min<size_t>((px + 255) / 256, 2048u * 8));
            hipLaunchKernelGGL(interleave_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->work2, dst + f * px * 3, static_cast<uint32_t>(px));
            HIP_TRY(ctx, hipGetLastError());
        }
        return BLUR_OK;
    }
    const size_t fstride = p.gen_gshift ? p.frame_elems : px * 3;
    for (int f = 0; f < nframes; ++f)
        if (int rc = launch_colpass<uint8_t, 3>(ctx, planes + f * fstride, dst + f * px * 3, rows, cols, p.sz.pad, *p.col, p.m_col, p.col_group, p.gen_gshift ? 1 : 0)) return rc;
    return BLUR_OK;
}

// ======================================================================================
// C ABI
// ======================================================================================
// ---- whole-image 2D path: helpers (the entry points are below, blur_pocketfft2d_*) ----
static int get_pos_of_freq(blur_ctx* ctx, DevicePlan* dp, const int** out)
{
    if (!dp->d_pos_of_freq) {
        const int n = dp->dev.n;
        std::vector<int> inv(n);
        for (int pos = 0; pos < n; ++pos) inv[dp->host.freq_of_pos[pos]] = pos;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dp->d_pos_of_freq), sizeof(int) * n));
        HIP_TRY(ctx, hipMemcpy(dp->d_pos_of_freq, inv.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    }
    *out = dp->d_pos_of_freq;
    return BLUR_OK;
}

template <bool CONV>
static int launch_img2d_cols(blur_ctx* ctx, float2* z, int s0, int s1, const DevicePlan& plan, const float* mcol, const float* krow_pos)
{
    // widest strip of columns whose lines fit the LDS: 8 columns = one 64-byte segment of a spectrum row
    const size_t line = static_cast<size_t>(line_stride(s0)) * sizeof(float2);
    int G = 8;
    while (G > 1 && G * line > kLdsLimit) G >>= 1;
    if (G * line > kLdsLimit) return fail(ctx, BLUR_ERR_UNSUPPORTED, "2D path: column FFT length exceeds LDS capacity");
    const int grid = (s1 + G - 1) / G;
#define BLUR_IMG2D_COLS(G_)                                                                                                        \
    case G_:                                                                                                                        \
        if (int rc = set_lds(ctx, img2d_cols_kernel<G_, CONV>, G_ * line)) return rc;                                              \
        hipLaunchKernelGGL((img2d_cols_kernel<G_, CONV>), dim3(grid), dim3(kThreads), G_ * line, ctx->stream, z, s0, s1, plan.dev, \
                           plan.d_tw, mcol, krow_pos);                                                                             \
        break;
    switch (G) {
        BLUR_IMG2D_COLS(8)
        BLUR_IMG2D_COLS(4)
        BLUR_IMG2D_COLS(2)
        BLUR_IMG2D_COLS(1)
    }
#undef BLUR_IMG2D_COLS
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

extern "C" {

void blur_opts_default(blur_opts* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->nyquist_quirk = 1;
}

int blur_gaussian_window(double sigma, int max_width) { return gaussian_window(sigma, max_width); }

int blur_get_gaussian(float* kernel, double sigma, int width, int fft_length)
{
    if (!kernel || !(sigma > 0) || width < 0 || fft_length < 0) return BLUR_ERR_INVALID;
    get_gaussian(kernel, sigma, width, fft_length);
    return BLUR_OK;
}

int blur_is_valid_size(int n) { return is_valid_size(n); }
int blur_nearest_transform_size(int n) { return nearest_transform_size(n); }

int blur_pffft_sizing(int rows, int cols, double sigma, int out[6])
{
    if (!out || rows <= 0 || cols <= 0 || !(sigma > 0)) return BLUR_ERR_INVALID;
    const Sizing s = pffft_sizing(rows, cols, sigma);
    out[0] = s.kSize; out[1] = s.pad; out[2] = s.n_col; out[3] = s.n_row; out[4] = s.tz_col; out[5] = s.tz_row;
    return BLUR_OK;
}

int blur_kernel_multipliers(double sigma, int ksize, int n, float* m)
{
    if (!m || !(sigma > 0) || ksize <= 0 || n < ksize) return BLUR_ERR_INVALID;
    kernel_multipliers(sigma, ksize, n, m);
    return BLUR_OK;
}

int blur_wr_kernel_multipliers(double sigma, int ksize, int n, int n_ref, int quirk, float* m)
{
    if (!m || !(sigma > 0) || ksize <= 0 || n < ksize || (n & 1) || n_ref <= 0) return BLUR_ERR_INVALID;
    std::vector<float> k(std::max(n, ksize));
    get_gaussian(k.data(), sigma, ksize, n);
    wr_multipliers(k.data(), n, n_ref, quirk != 0, m);
    return BLUR_OK;
}

int blur_fft_plan_radices(int n, int* radices)
{
    FftPlan p;
    if (!radices || !make_plan(n, p)) return 0;
    for (int i = 0; i < p.npass; ++i) radices[i] = p.radix[i];
    return p.npass;
}

int blur_ctx_create(blur_ctx** out, int device)
{
    if (!out) return BLUR_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return BLUR_ERR_HIP;
    }
    if (device < 0 || device >= count) { g_create_err = "device ordinal out of range"; return BLUR_ERR_INVALID; }
    e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    blur_ctx* c = new blur_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    int xccs = 0;   // the XCD-banded task maps of the matrix-core kernels (fx_kernels.hpp, mx_kernels.hpp) take the count from the device
    if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && xccs > 0) c->num_xcds = xccs;
    *out = c;
    return BLUR_OK;
}

int blur_ctx_destroy(blur_ctx* ctx)
{
    if (!ctx) return BLUR_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->plans) {
        if (kv.second->d_tw) (void)hipFree(kv.second->d_tw);
        if (kv.second->d_pos_of_freq) (void)hipFree(kv.second->d_pos_of_freq);
    }
    for (auto& kv : ctx->spectra) (void)hipFree(kv.second);
    if (ctx->d_w256) (void)hipFree(ctx->d_w256);
    for (auto& kv : ctx->wr_tw0) (void)hipFree(kv.second);
    for (auto& kv : ctx->wr_spectra) (void)hipFree(kv.second);
    for (auto& kv : ctx->mx_tables) {
        (void)hipFree(kv.second.frags_row); (void)hipFree(kv.second.frags_col);
        (void)hipFree(kv.second.taps_row);
    }
    for (auto& lt : ctx->lines_tables) (void)hipFree(lt.dev);
    if (ctx->fx_strips) (void)hipFree(ctx->fx_strips);
    if (ctx->fx_sums) (void)hipFree(ctx->fx_sums);
    if (ctx->tl_sums) (void)hipFree(ctx->tl_sums);
    if (ctx->tl_terms) (void)hipFree(ctx->tl_terms);
    if (ctx->tl_copy) (void)hipFree(ctx->tl_copy);
    for (auto& kv : ctx->tl_taps) (void)hipFree(kv.second);
    if (ctx->mx_sums) (void)hipFree(ctx->mx_sums);
    if (ctx->mx_terms) (void)hipFree(ctx->mx_terms);
    if (ctx->work) (void)hipFree(reinterpret_cast<char*>(ctx->work) - kWorkGuard);
    if (ctx->work2) (void)hipFree(ctx->work2);
    if (ctx->box_tmp) (void)hipFree(ctx->box_tmp);
    if (ctx->host_stage) (void)hipFree(ctx->host_stage);
    if (ctx->pipe.ready) {
        (void)hipStreamSynchronize(ctx->pipe.h2d);
        (void)hipStreamSynchronize(ctx->pipe.d2h);
        for (int k = 0; k < HostPipe::S; ++k) {
            if (ctx->pipe.buf[k]) (void)hipFree(ctx->pipe.buf[k]);
            (void)hipEventDestroy(ctx->pipe.in_done[k]);
            (void)hipEventDestroy(ctx->pipe.comp_done[k]);
            (void)hipEventDestroy(ctx->pipe.out_done[k]);
        }
        (void)hipStreamDestroy(ctx->pipe.h2d);
        (void)hipStreamDestroy(ctx->pipe.d2h);
    }
    for (auto& t : ctx->ev_busy) { (void)hipEventDestroy(std::get<0>(t)); (void)hipEventDestroy(std::get<1>(t)); }
    for (auto& t : ctx->ev_free) { (void)hipEventDestroy(t.first); (void)hipEventDestroy(t.second); }
    delete ctx;
    return BLUR_OK;
}

int blur_ctx_set_stream(blur_ctx* ctx, void* hip_stream)
{
    if (!ctx) return BLUR_ERR_INVALID;
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (next != ctx->stream) {
        // the workspace and the cached tables are shared: work queued on the old stream must be
        // finished before another stream may touch them
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->stream = next;
    }
    return BLUR_OK;
}

int blur_ctx_synchronize(blur_ctx* ctx)
{
    if (!ctx) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BLUR_OK;
}

const char* blur_last_error(const blur_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int blur_ctx_timing_enable(blur_ctx* ctx, int on)
{
    if (!ctx) return BLUR_ERR_INVALID;
    ctx->timing = on == 2 ? 2 : on != 0 ? 1 : 0;
    return BLUR_OK;
}

int blur_ctx_timing(blur_ctx* ctx, double out_ms[2], int out_launches[2], int out_frames[2], int reset)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (int rc = timing_drain(ctx)) return rc;
    for (int i = 0; i < 2; ++i) {
        if (out_ms) out_ms[i] = ctx->ms[i];
        if (out_launches) out_launches[i] = ctx->launches[i];
        if (out_frames) out_frames[i] = ctx->frames[i];
        if (reset) { ctx->ms[i] = 0; ctx->launches[i] = 0; ctx->frames[i] = 0; }
    }
    return BLUR_OK;
}

// both passes on the matrix cores; V (24-bit fixed point, mx_kernels.hpp) lives in the float workspace
static int run_mx_u8c3(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int nframes, int rows, int cols, int chunk, const Prepared& p)
{
    const size_t px = static_cast<size_t>(rows) * cols;
    MxGeom g{ rows, cols, p.sz.pad, p.mx_vpitch, 0, 0, mx_vrows(rows, p.mx->nkb), ctx->num_xcds };
    const int chunks = (cols + kMxRowChunk - 1) / kMxRowChunk, rblocks = g.vrows / 32, zblocks = (g.vrows + 255) / 256;
    // the quirk's scratch per frame: spart int [chunks][rows][3]; then floats: vpart [rblocks][vpitch], qrow [3][vrows], qcol [vpitch];
    // then doubles: zpart [zblocks][3]
    const size_t n_spart = static_cast<size_t>(chunks) * rows * 3, n_vpart = static_cast<size_t>(rblocks) * g.vpitch,
                 n_qrow = static_cast<size_t>(3) * g.vrows, n_qcol = g.vpitch, n_zpart = static_cast<size_t>(zblocks) * 3;
    const size_t sums_bytes = n_spart * sizeof(int) * chunk;
    const size_t terms_bytes = ((n_vpart + n_qrow + n_qcol) * sizeof(float) + n_zpart * sizeof(double)) * chunk + 64;
    if (p.mx_quirk) {
        if (ctx->mx_sums_bytes < sums_bytes) {
            if (ctx->mx_sums) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->mx_sums)); ctx->mx_sums = nullptr; ctx->mx_sums_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->mx_sums), sums_bytes));
            ctx->mx_sums_bytes = sums_bytes;
        }
        if (ctx->mx_terms_bytes < terms_bytes) {
            if (ctx->mx_terms) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->mx_terms)); ctx->mx_terms = nullptr; ctx->mx_terms_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->mx_terms), terms_bytes));
            ctx->mx_terms_bytes = terms_bytes;
        }
    }
    for (int f = 0; f < nframes; f += chunk) {
        const int nf = nframes - f < chunk ? nframes - f : chunk;
        const uint8_t* s = d_src + static_cast<size_t>(f) * px * 3;
        uint8_t* d = d_dst + static_cast<size_t>(f) * px * 3;
        g.nframes = nf;
        g.aligned = ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(s) & 3) == 0) ? 1 : 0;
        int* spart = nullptr;
        float *vpart = nullptr, *qrow = nullptr, *qcol = nullptr;
        double* zpart = nullptr;
        if (p.mx_quirk) {
            spart = ctx->mx_sums;
            zpart = reinterpret_cast<double*>(ctx->mx_terms);                      // doubles first: 8-byte aligned
            vpart = reinterpret_cast<float*>(zpart + n_zpart * nf);
            qrow = vpart + n_vpart * nf;
            qcol = qrow + n_qrow * nf;
        }
        { TimedLaunch t(ctx, 0, nf);
          HIP_TRY(ctx, p.mx->row_u8(ctx->stream, s, ctx->work, p.mxt->frags_row, g, ctx->num_cus, spart, vpart)); }
        if (p.mx_quirk) {
            // the Nyquist-slot terms from the row kernel's partial sums: two small launches between the passes
            hipLaunchKernelGGL(mx_quirk_rows, dim3(zblocks, nf), dim3(256), 0, ctx->stream, spart, qrow, zpart, g, chunks, 8 * (p.mx->nkb - 2), p.mxt->dr);
            HIP_TRY(ctx, hipGetLastError());
            hipLaunchKernelGGL(mx_quirk_cols, dim3((g.vpitch + 255) / 256, nf), dim3(256), 0, ctx->stream, vpart, zpart, qcol, g, rblocks, zblocks, p.mxt->dr, p.mxt->dc);
            HIP_TRY(ctx, hipGetLastError());
        }
        { TimedLaunch t(ctx, 1, nf);
          HIP_TRY(ctx, p.mx->col_u8(ctx->stream, ctx->work, d, p.mxt->frags_col, g, qcol, ctx->num_cus, qrow)); }
    }
    return BLUR_OK;
}

// both passes in one kernel on the matrix cores (fx_kernels.hpp): no intermediate in memory
static int run_fx_u8c3(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int nframes, int rows, int cols, const Prepared& p, float* vdump = nullptr)
{
    const size_t px = static_cast<size_t>(rows) * cols;
    // in place (the reference's own calling convention, Source.cpp:429,567): a strip reads its neighbours' columns and the rows
    // below while they are being written, so the frames are read from a copy in the workspace
    const uint8_t* lo = d_src < d_dst ? d_src : d_dst;
    const uint8_t* hi = d_src < d_dst ? d_dst : d_src;
    if (static_cast<size_t>(hi - lo) < px * 3 * nframes) {
        // (a large batch: in parts of at most 1 GiB, the cap of every engine's workspace -- frames are disjoint, so a part's result
        // never touches a later part's source)
        const size_t cap = std::max<size_t>(1, (static_cast<size_t>(1) << 30) / (px * 3));
        if (d_src == d_dst && static_cast<size_t>(nframes) > cap) {
            for (int f0 = 0; f0 < nframes; f0 += static_cast<int>(cap)) {
                const int nf = std::min<int>(static_cast<int>(cap), nframes - f0);
                if (int rc = run_fx_u8c3(ctx, d_src + static_cast<size_t>(f0) * px * 3, d_dst + static_cast<size_t>(f0) * px * 3, nf, rows, cols, p, vdump)) return rc;
            }
            return BLUR_OK;
        }
        if (int rc = ensure_work(ctx, px * 3 * nframes)) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->work, d_src, px * 3 * nframes, hipMemcpyDeviceToDevice, ctx->stream));
        d_src = reinterpret_cast<const uint8_t*>(ctx->work);
    }
    const int nkb = p.fx->nkb, pada = 8 * (nkb - 2);
    FxGeom g{ rows, cols, p.sz.pad, nframes, 0, (rows + 31) / 32, fx_right_strips(cols, pada), ctx->num_xcds };
    g.aligned = ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(d_src) & 3) == 0 && (reinterpret_cast<uintptr_t>(d_dst) & 3) == 0) ? 1 : 0;
    {   // the edge chunks' windows
        const int win = kFxChunk + 2 * pada, nstrips = fx_left_strips(pada) + g.nright;
        const size_t bytes = static_cast<size_t>(nframes) * nstrips * rows * win * 3 + 64;
        if (ctx->fx_strips_bytes < bytes) {
            if (ctx->fx_strips) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->fx_strips)); ctx->fx_strips = nullptr; ctx->fx_strips_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->fx_strips), bytes));
            ctx->fx_strips_bytes = bytes;
        }
    }
    // (the three-channel kernels read only the loads that touch a mirrored pixel from their strips: fx_strip_range)
    const int chunks_x = (cols + kFxChunk - 1) / kFxChunk, narrow =
#ifdef FX_FULL_STRIPS
        0;
#else
        p.fx->nkb <= 11 ? 1 : 0;
#endif
    int strip_groups = (kFxChunk + 2 * pada) / 4;
    if (narrow) {
        const int gpr = strip_groups, per = (gpr + 7) / 8, nleft = fx_left_strips(pada);
        strip_groups = 0;
        for (int si = 0; si < nleft + g.nright; ++si) {
            int lo, hi;
            fx_strip_range(si < nleft ? si : chunks_x - g.nright + si - nleft, cols, pada, per, &lo, &hi);
            strip_groups = std::max(strip_groups, std::min(8 * hi, gpr) - 8 * lo);
        }
        if (strip_groups < 1) strip_groups = 1;
    }
    const int strip_blocks = (((rows + 3) / 4) * strip_groups + 255) / 256;
    const int n_strip = strip_blocks * (fx_left_strips(pada) + g.nright) * nframes;
    FxQuirk qk{};
    if (!p.mx_quirk) {
        TimedLaunch t(ctx, 1, nframes);
        hipLaunchKernelGGL(fx_prepass<1>, dim3(n_strip), dim3(256), 0, ctx->stream, d_src, nullptr, nullptr, nullptr, ctx->fx_strips, rows, cols, p.sz.pad, pada, 1, 1, 0, chunks_x,
                           g.nright, strip_blocks, kFxSumRows, narrow);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (p.mx_quirk) {
        // the quirk's partial sums (exact integers): srow_part [frame][batch][row][3], cpart [frame][band][3 cols], zpart (64-bit)
        // [frame][band][batch][3]; the fused kernel adds them up where it needs them (struct FxQuirk)
        const int gpt = fx_groups_per_thread(cols);
        const int groups = (cols + 3) / 4;
        const int band_rows = fx_band_rows(rows, cols, nframes, ctx->num_cus), nbands = (rows + band_rows - 1) / band_rows, nbatches = (groups + 256 * gpt - 1) / (256 * gpt);
        auto up4 = [](size_t v) { return (v + 3) & ~static_cast<size_t>(3); };
        const size_t n_srow = up4(static_cast<size_t>(nframes) * nbatches * rows * 3), n_cpart = up4(static_cast<size_t>(nframes) * nbands * 12 * groups);
        const size_t n_z = static_cast<size_t>(nframes) * nbands * nbatches * 3;
        const size_t bytes = (n_srow + n_cpart) * sizeof(int) + n_z * sizeof(long long) + 64;
        if (ctx->fx_sums_bytes < bytes) {
            if (ctx->fx_sums) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->fx_sums)); ctx->fx_sums = nullptr; ctx->fx_sums_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->fx_sums), bytes + bytes / 8));
            ctx->fx_sums_bytes = bytes + bytes / 8;
        }
        int* srow = ctx->fx_sums;
        int* cpart = srow + n_srow;                                                     // 16-byte aligned: int4 stores
        long long* zpart = reinterpret_cast<long long*>(cpart + n_cpart);
        { TimedLaunch t(ctx, 1, nframes);
          const int n_alt = nbands * nbatches * nframes;
          auto kern = gpt == 1 ? fx_prepass<1> : (gpt == 2 ? fx_prepass<2> : fx_prepass<4>);
          hipLaunchKernelGGL(kern, dim3(n_alt + n_strip), dim3(256), 0, ctx->stream, d_src, srow, cpart, zpart, ctx->fx_strips, rows, cols, p.sz.pad, pada, nbands,
                             nbatches, n_alt, chunks_x, g.nright, strip_blocks, band_rows, narrow);
          HIP_TRY(ctx, hipGetLastError()); }
        qk.srow_part = srow;
        qk.cpart = cpart;
        qk.zpart = zpart;
        qk.taps = p.mxt->taps_row;
        qk.nbatches = nbatches;
        qk.nbands = nbands;
        qk.cpitch = 12 * groups;
        qk.dr = p.mxt->dr;
        qk.dc = p.mxt->dc;
    }
    // diagnostic (timing-only -DFX_STAMPS builds of the kernel write here; see tools/fx_variants.sh): cycles per phase kind
    unsigned long long* stamps = nullptr;
    if (!p.mx_quirk && !vdump && std::getenv("BLUR_FX_STAMPS")) {
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&stamps), 64));
        HIP_TRY(ctx, hipMemset(stamps, 0, 64));
    }
    { TimedLaunch t(ctx, 0, nframes);
      HIP_TRY(ctx, p.fx->blur_u8(ctx->stream, d_src, d_dst, p.mxt->frags_row, g, ctx->num_cus, p.mx_quirk ? &qk : nullptr, ctx->fx_strips, vdump, stamps)); }
    if (stamps) {
        unsigned long long h[8];
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(h, stamps, 64, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipFree(stamps));
        if (h[7]) {
            std::fprintf(stderr, "fx stamps: steps %llu, cycles per step: A0 %.0f B0 %.0f A1 %.0f B1 %.0f A2 %.0f B2 %.0f, whole task %.0f per step\n", h[7], h[0] / double(h[7]),
                         h[1] / double(h[7]), h[2] / double(h[7]), h[3] / double(h[7]), h[4] / double(h[7]), h[5] / double(h[7]), h[6] / double(h[7]));
        }
    }
    return BLUR_OK;
}

static bool bands_overlap(const uint8_t* lo, const uint8_t* hi, size_t bytes) { return static_cast<size_t>(hi - lo) < bytes; }

// Tiled wave-resident path (prepare(): plan_tiled): frame by frame, the quirk's sums (fx_prepass) and terms (tl_terms_kernel), then
// per band of rows the column kernel and per tile of columns the row kernel.  Intermediate: one band at a time in the workspace.
static int run_wr_tiled(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int nframes, int rows, int cols, const Prepared& p)
{
    const size_t px = static_cast<size_t>(rows) * cols, fb = px * 3;
    const int pad = p.sz.pad;
    // (the pre-pass with one group of 4 pixels per thread: the term kernel adds up any number of batch parts, unlike the fused kernel)
    const int groups = (cols + 3) / 4, gpt = 1;
    if (p.tl_quirk && cols < 4) return fail(ctx, BLUR_ERR_UNSUPPORTED, "tiled wave-resident path: image narrower than 4 pixels (the quirk's pre-pass)");
    const int pitch = (cols + 7) & ~7;
    int *srow = nullptr, *cpart = nullptr;
    long long* zpart = nullptr;
    float *e = nullptr, *h = nullptr;
    int band_rows = 0, nbands = 0, nbatches = 0;
    if (p.tl_quirk) {
        // bands as tall as still give four workgroups per CU: the term kernel adds up the bands' parts per column
        band_rows = static_cast<long long>(rows) * cols < 4000000ll ? 16 : kFxSumRows;
        while (band_rows < 256 && static_cast<long long>((groups + 255) / 256) * ((rows + 2 * band_rows - 1) / (2 * band_rows)) >= 4ll * ctx->num_cus) band_rows *= 2;
        nbands = (rows + band_rows - 1) / band_rows;
        nbatches = (groups + 256 * gpt - 1) / (256 * gpt);
        auto up4 = [](size_t v) { return (v + 3) & ~static_cast<size_t>(3); };
        const size_t n_srow = up4(static_cast<size_t>(nbatches) * rows * 3), n_cpart = up4(static_cast<size_t>(nbands) * 12 * groups);
        const size_t n_z = static_cast<size_t>(nbands) * nbatches * 3;
        const size_t bytes = (n_srow + n_cpart) * sizeof(int) + n_z * sizeof(long long) + 64;
        if (ctx->tl_sums_bytes < bytes) {
            if (ctx->tl_sums) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->tl_sums)); ctx->tl_sums = nullptr; ctx->tl_sums_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->tl_sums), bytes));
            ctx->tl_sums_bytes = bytes;
        }
        srow = ctx->tl_sums;
        cpart = srow + n_srow;
        zpart = reinterpret_cast<long long*>(cpart + n_cpart);
        const size_t tbytes = (static_cast<size_t>(3) * pitch + static_cast<size_t>(3) * rows) * sizeof(float) + 64;
        if (ctx->tl_terms_bytes < tbytes) {
            if (ctx->tl_terms) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->tl_terms)); ctx->tl_terms = nullptr; ctx->tl_terms_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->tl_terms), tbytes));
            ctx->tl_terms_bytes = tbytes;
        }
        e = ctx->tl_terms;
        h = e + static_cast<size_t>(3) * pitch;
    }
    for (int f = 0; f < nframes; ++f) {
        const uint8_t* src = d_src + static_cast<size_t>(f) * fb;
        uint8_t* dst = d_dst + static_cast<size_t>(f) * fb;
        // in place (or overlapping): a band reads rows the band before it has already written
        const uint8_t* lo = src < dst ? src : dst;
        const uint8_t* hi = src < dst ? dst : src;
        if (bands_overlap(lo, hi, fb) && (p.bands.size() > 1 || p.tiles.size() > 1 || src == dst)) {
            if (ctx->tl_copy_bytes < fb) {
                if (ctx->tl_copy) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->tl_copy)); ctx->tl_copy = nullptr; ctx->tl_copy_bytes = 0; }
                HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->tl_copy), fb));
                ctx->tl_copy_bytes = fb;
            }
            HIP_TRY(ctx, hipMemcpyAsync(ctx->tl_copy, src, fb, hipMemcpyDeviceToDevice, ctx->stream));
            src = ctx->tl_copy;
        }
        if (p.tl_quirk) {
            TimedLaunch t(ctx, 1, 1);
            const int n_alt = nbands * nbatches;
            hipLaunchKernelGGL(fx_prepass<1>, dim3(n_alt), dim3(256), 0, ctx->stream, src, srow, cpart, zpart, nullptr, rows, cols, pad, 0, nbands, nbatches, n_alt, 1, 0, 1, band_rows, 0);
            HIP_TRY(ctx, hipGetLastError());
            const int ne = (pitch + 255) / 256, nh = (rows + 255) / 256;
            const size_t lds = (static_cast<size_t>(3) * (256 + 2 * pad) + (2 * pad + 1) + 3 + 3 * 256) * sizeof(double);
            if (lds > 64 * 1024) HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tl_terms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            hipLaunchKernelGGL(tl_terms_kernel, dim3(ne + nh, 1), dim3(256), lds, ctx->stream, srow, cpart, zpart, p.tl_taps, e, h, rows, cols, pad, pitch, nbatches, nbands,
                               12 * groups, p.tl_dr, p.tl_dc, ne);
            HIP_TRY(ctx, hipGetLastError());
        }
        // consecutive middle bands of one shape, a constant (even) number of rows apart, are the "frames" of ONE launch per kernel
        auto group_of = [&](size_t bi, int& step) -> int {
            const Prepared::Span& b = p.bands[bi];
            const int brows = b.in1 - b.in0;
            int nb = 1;
            step = 0;
            if (b.padmode == 0) {
                while (bi + nb < p.bands.size()) {
                    const Prepared::Span& c = p.bands[bi + nb];
                    const int st = c.in0 - p.bands[bi + nb - 1].in0;
                    if (c.padmode != 0 || c.in1 - c.in0 != brows || c.v0 - c.in0 != b.v0 - b.in0 || c.v1 - c.v0 != b.v1 - b.v0 || (st & 1) || (nb > 1 && st != step)) break;
                    step = st;
                    ++nb;
                }
            }
            if (nb > 1 && wr_frame_floats(brows, cols, b.padmode, p.tl_col->g) * nb * sizeof(float) > (static_cast<size_t>(3) << 30)) { nb = 1; step = 0; }
            return nb;
        };
        {
            size_t need = 0;
            for (size_t bi = 0; bi < p.bands.size();) {
                int step;
                const int nb = group_of(bi, step);
                need = std::max(need, wr_frame_floats(p.bands[bi].in1 - p.bands[bi].in0, cols, p.bands[bi].padmode, p.tl_col->g) * nb * sizeof(float));
                bi += nb;
            }
            if (int rc = ensure_work(ctx, need)) return rc;
        }
        float* const inter = reinterpret_cast<float*>(ctx->work);
        for (size_t bi = 0; bi < p.bands.size();) {
            const Prepared::Span& b = p.bands[bi];
            const int brows = b.in1 - b.in0;
            int step = 0;
            const int nb = group_of(bi, step);
            const int vr_max = bi + nb < p.bands.size() ? p.bands[bi + nb].v0 : rows;       // rows from here on are the next band's
            WrColTerm term;
            if (p.tl_quirk) {
                term.e = e;
                term.pitch = pitch;
                term.sign = ((b.in0 + pad - b.padmode) & 1) ? -1.f : 1.f;       // (-1)^(image row + pad) of slot 0: slot = (band row + padmode) & 1
            }
            term.band_step = nb > 1 ? step : 0;
            { TimedLaunch t(ctx, 1, 1);
              HIP_TRY(ctx, p.tl_col->col_u8(ctx->stream, src + static_cast<size_t>(b.in0) * cols * 3, inter, brows, cols, b.padmode, nb, ctx->num_cus, ctx->d_w256, p.tl_tw0_col,
                                            p.tl_m_col, term)); }
            const int g = p.tl_col->g, npairs = wr_npairs(brows, b.padmode), strips_full = (cols + g - 1) / g;
            for (size_t ti = 0; ti < p.tiles.size(); ++ti) {
                const Prepared::Span& t = p.tiles[ti];
                WrRowTile tile;
                tile.h = p.tl_quirk ? h : nullptr;
                tile.rows_full = rows;
                tile.row_base = b.in0;
                tile.vr0 = b.v0;
                tile.vr1 = b.v1;
                tile.ypar = b.padmode & 1;
                tile.dst_pitch = cols * 3;
                tile.dst_x0 = t.in0;
                tile.vc0 = t.v0 - t.in0;
                tile.vc1 = t.v1 - t.in0;
                tile.xpar = (t.in0 + pad) & 1;                                   // (-1)^(image column + pad): the call's column 0 is image column in0
                tile.plane_strips = strips_full;
                tile.g = g;
                tile.t0 = (b.v0 - b.in0 + tile.ypar) >> 1;                        // the pairs that hold a kept row
                tile.tn = ((b.v1 - 1 - b.in0 + tile.ypar) >> 1) - tile.t0 + 1;
                tile.band_step = nb > 1 ? step : 0;
                tile.vr_max = vr_max;
                if (nb == 1 && tile.vr1 > vr_max) tile.vr1 = vr_max;
                TimedLaunch tl(ctx, 0, 1);
                HIP_TRY(ctx, p.tl_row[ti]->row_u8(ctx->stream, inter + static_cast<size_t>(t.in0 / g) * npairs * (2 * g), dst, brows, t.in1 - t.in0, t.padmode, nb, ctx->num_cus,
                                                  ctx->d_w256, p.tl_tw0_row[ti], p.tl_m_row[ti], tile));
            }
            bi += nb;
        }
    }
    return BLUR_OK;
}

static int blur_u8c3_batch_impl(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int nframes,
                                int rows, int cols, double sigma, const blur_opts* opts, const CustomKernel* ck)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_src || !d_dst || nframes < 0) return fail(ctx, BLUR_ERR_INVALID, "null frame pointer or negative frame count");
    Prepared p;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 3) == 0;
    if (int rc = prepare(ctx, rows, cols, sigma, opts, p, true, ck, true, aligned)) return rc;
    const size_t px = static_cast<size_t>(rows) * cols;
    // Frames per launch pair.  Measured on MI355X (4K, sigma 20): 1 frame 0.170 ms/frame, 2: 0.148, 4: 0.143,
    // 8: 0.140 -- the kernels are latency bound, not HBM bound, so filling every CU evenly and amortising
    // the per-workgroup table loads is worth more than keeping the float intermediate inside the 256 MiB
    // Infinity Cache.  The workspace is capped at 1 GiB.
    if (p.fx) {
        if (nframes == 0) return BLUR_OK;
        ctx->last_family = 6;
        return run_fx_u8c3(ctx, d_src, d_dst, nframes, rows, cols, p);
    }
    if (p.tiled) {
        if (nframes == 0) return BLUR_OK;
        ctx->last_family = 7;
        return run_wr_tiled(ctx, d_src, d_dst, nframes, rows, cols, p);
    }
    int chunk = opts && opts->frames_per_launch > 0 ? opts->frames_per_launch : static_cast<int>((1024u << 20) / (p.frame_elems * sizeof(float)));
    if (chunk < 1) chunk = 1;
    if (chunk > nframes) chunk = nframes;
    if (nframes == 0) return BLUR_OK;
    if (int rc = ensure_work(ctx, p.frame_elems * sizeof(float) * chunk)) return rc;
    ctx->last_family = p.mx ? 4 : p.wr_col ? 2 : ((p.row->fast && p.col->fast) ? 1 : 0);
    if (p.mx) return run_mx_u8c3(ctx, d_src, d_dst, nframes, rows, cols, chunk, p);
    for (int f = 0; f < nframes; f += chunk) {
        const int nf = nframes - f < chunk ? nframes - f : chunk;
        const uint8_t* s = d_src + static_cast<size_t>(f) * px * 3;
        uint8_t* d = d_dst + static_cast<size_t>(f) * px * 3;
        if (p.wr_col) {
            // columns first, then rows (wr_kernels.hpp); timing slot 1 = column kernel, 0 = row kernel as elsewhere
            { TimedLaunch t(ctx, 1, nf);
              HIP_TRY(ctx, p.wr_col->col_u8(ctx->stream, s, ctx->work, rows, cols, p.sz.pad, nf, ctx->num_cus, ctx->d_w256, p.wr_tw0_col, p.wr_m_col, WrColTerm{})); }
            { TimedLaunch t(ctx, 0, nf);
              WrRowTile whole;
              whole.g = p.wr_col->g;
              HIP_TRY(ctx, p.wr_row->row_u8(ctx->stream, ctx->work, d, rows, cols, p.sz.pad, nf, ctx->num_cus, ctx->d_w256, p.wr_tw0_row, p.wr_m_row, whole)); }
            continue;
        }
        if (int rc = run_rowpass_u8c3(ctx, s, ctx->work, rows, cols, nf, p, p.tile_w)) return rc;
        if (int rc = run_colpass_u8c3(ctx, ctx->work, d, rows, cols, nf, p)) return rc;
    }
    return BLUR_OK;
}

int blur_gaussian_u8c3_batch_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int nframes,
                                 int rows, int cols, double sigma, const blur_opts* opts)
{
    return blur_u8c3_batch_impl(ctx, d_src, d_dst, nframes, rows, cols, sigma, opts, nullptr);
}

int blur_separable_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols,
                            const float* taps, int ksize, int pad, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!taps || ksize <= 0 || (ksize & 1) == 0 || pad < 0) return fail(ctx, BLUR_ERR_INVALID, "separable kernel: odd tap count and pad >= 0 required");
    for (int t = 0; t < ksize / 2; ++t)
        if (taps[t] != taps[ksize - 1 - t]) return fail(ctx, BLUR_ERR_UNSUPPORTED, "separable kernel must be symmetric (real spectrum, Source.cpp:419)");
    CustomKernel ck;
    ck.taps = taps; ck.ksize = ksize; ck.pad = pad;
    return blur_u8c3_batch_impl(ctx, d_src, d_dst, 1, rows, cols, 0., opts, &ck);
}

int blur_boxfft_sizing(int rows, int cols, double nsmooth, int out[4])
{
    if (!out || rows <= 1 || cols <= 1 || !(nsmooth >= 1)) return BLUR_ERR_INVALID;
    int klen, pad;
    boxfft_sizing(rows, cols, nsmooth, klen, pad);
    const int n0 = rows + 2 * pad, n1 = cols + 2 * pad;
    out[0] = klen; out[1] = pad;
    out[2] = is_valid_size(n0) ? n0 : nearest_transform_size(n0);
    out[3] = is_valid_size(n1) ? n1 : nearest_transform_size(n1);
    return BLUR_OK;
}

int blur_box_kernel(float* kernel, int klen, int fft_length)
{
    if (!kernel || klen <= 0 || fft_length < 2 * klen + 2) return BLUR_ERR_INVALID;
    std::fill(kernel, kernel + fft_length, 0.f);
    box_kernel_1d(kernel, klen, fft_length);
    return BLUR_OK;
}

int blur_boxfft_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols, double nsmooth, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (rows <= 1 || cols <= 1 || !(nsmooth >= 1)) return fail(ctx, BLUR_ERR_INVALID, "boxfft: rows, cols > 1 and nsmooth >= 1 required");
    CustomKernel ck;
    boxfft_sizing(rows, cols, nsmooth, ck.box_klen, ck.pad);
    if (ck.box_klen < 1) return fail(ctx, BLUR_ERR_INVALID, "boxfft: empty kernel");
    return blur_u8c3_batch_impl(ctx, d_src, d_dst, 1, rows, cols, 0., opts, &ck);
}

int blur_gaussian_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols, double sigma, const blur_opts* opts)
{
    return blur_gaussian_u8c3_batch_dev(ctx, d_src, d_dst, 1, rows, cols, sigma, opts);
}

int blur_gaussian_f32c1_dev(blur_ctx* ctx, const float* d_src, float* d_dst, int rows, int cols, double sigma, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_src || !d_dst) return fail(ctx, BLUR_ERR_INVALID, "null plane pointer");
    Prepared p;
    if (int rc = prepare(ctx, rows, cols, sigma, opts, p, false)) return rc;
    const size_t px = static_cast<size_t>(rows) * cols;
    if (int rc = ensure_work(ctx, px * sizeof(float))) return rc;
    if (int rc = launch_rowpass<float, 1>(ctx, d_src, ctx->work, rows, cols, p.sz.pad, *p.row, p.m_row)) return rc;
    return launch_colpass<float, 1>(ctx, ctx->work, d_dst, rows, cols, p.sz.pad, *p.col, p.m_col, p.col_group);
}

int blur_rowpass_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, float* d_planes, int rows, int cols, double sigma, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_src || !d_planes) return fail(ctx, BLUR_ERR_INVALID, "null pointer");
    Prepared p;
    if (opts && opts->engine == BLUR_ENGINE_FUSED) {
        // the fused kernel's row-pass planes (tests): the whole blur runs, its bytes go to the workspace and are dropped
        const bool aligned = (reinterpret_cast<uintptr_t>(d_src) & 3) == 0;
        if (int rc = prepare(ctx, rows, cols, sigma, opts, p, true, nullptr, true, aligned)) return rc;
        if (!p.fx || p.fx->nkb > 11 || (cols & 3) != 0)
            return fail(ctx, BLUR_ERR_UNSUPPORTED, "row-pass planes of the fused engine: a test instantiation for pad <= 72 and widths that are multiples of 4");
        const size_t bytes = static_cast<size_t>(rows) * cols * 3;
        if (int rc = ensure_work(ctx, 2 * bytes + 64)) return rc;
        return run_fx_u8c3(ctx, d_src, reinterpret_cast<uint8_t*>(ctx->work) + ((bytes + 63) & ~static_cast<size_t>(63)), 1, rows, cols, p, d_planes);
    }
    if (int rc = prepare(ctx, rows, cols, sigma, opts, p, true, nullptr, false)) return rc;   // the rows-first kernels: row-major planes
    return run_rowpass_u8c3(ctx, d_src, d_planes, rows, cols, 1, p, -1);   // always row-major for the caller
}

int blur_gaussian_u8c3_host(blur_ctx* ctx, const uint8_t* src, uint8_t* dst, int rows, int cols, double sigma, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0) return fail(ctx, BLUR_ERR_INVALID, "null image or non-positive size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(rows) * cols * 3, half = (bytes + 255) & ~static_cast<size_t>(255);
    void* dv = nullptr;
    // two device images, source and destination: the fused engine reads its neighbours' pixels while it writes and would otherwise
    // copy an in-place frame into its workspace first
    if (int rc0 = ensure_host_stage(ctx, 2 * half, &dv)) return rc0;
    uint8_t* d = static_cast<uint8_t*>(dv);
    int rc = BLUR_OK;
    hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = blur_gaussian_u8c3_dev(ctx, d, d + half, rows, cols, sigma, opts);
    if (e == hipSuccess && rc == BLUR_OK) e = hipMemcpyAsync(dst, d + half, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("host blur: ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    return rc;
}

int blur_gaussian_u8c3_host_batch(blur_ctx* ctx, const uint8_t* src, uint8_t* dst, int nframes, int rows, int cols, double sigma,
                                  const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!src || !dst || nframes < 0 || rows <= 0 || cols <= 0) return fail(ctx, BLUR_ERR_INVALID, "null image, negative count or non-positive size");
    if (nframes == 0) return BLUR_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t fb = static_cast<size_t>(rows) * cols * 3;
    HostPipe& p = ctx->pipe;
    constexpr int S = HostPipe::S;
    if (!p.ready) {
        // non-blocking: the context's stream may be the null stream, which would otherwise serialise with these
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p.h2d, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p.d2h, hipStreamNonBlocking));
        for (int k = 0; k < S; ++k) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&p.in_done[k], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&p.comp_done[k], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&p.out_done[k], hipEventDisableTiming));
        }
        p.ready = true;
    }
    if (p.bytes < fb) {
        for (int k = 0; k < S; ++k) {
            if (p.buf[k]) { HIP_TRY(ctx, hipFree(p.buf[k])); p.buf[k] = nullptr; }
        }
        p.bytes = 0;
        for (int k = 0; k < S; ++k) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&p.buf[k]), 2 * ((fb + 255) & ~static_cast<size_t>(255))));   // input | output
        p.bytes = fb;
    }
    int rc = BLUR_OK;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    for (int i = 0; i < nframes && e == hipSuccess && rc == BLUR_OK; ++i) {
        const int k = i % S;
        if (i >= S && !step(hipStreamWaitEvent(p.h2d, p.out_done[k], 0))) break;   // the slot's previous frame has left
        if (!step(hipMemcpyAsync(p.buf[k], src + static_cast<size_t>(i) * fb, fb, hipMemcpyHostToDevice, p.h2d))) break;
        if (!step(hipEventRecord(p.in_done[k], p.h2d))) break;
        if (!step(hipStreamWaitEvent(ctx->stream, p.in_done[k], 0))) break;
        uint8_t* const obuf = p.buf[k] + ((fb + 255) & ~static_cast<size_t>(255));           // out of place: no copy into the workspace
        rc = blur_gaussian_u8c3_dev(ctx, p.buf[k], obuf, rows, cols, sigma, opts);
        if (rc != BLUR_OK) break;
        if (!step(hipEventRecord(p.comp_done[k], ctx->stream))) break;
        if (!step(hipStreamWaitEvent(p.d2h, p.comp_done[k], 0))) break;
        if (!step(hipMemcpyAsync(dst + static_cast<size_t>(i) * fb, obuf, fb, hipMemcpyDeviceToHost, p.d2h))) break;
        if (!step(hipEventRecord(p.out_done[k], p.d2h))) break;
    }
    // drain everything that was queued, whatever happened above
    const std::string first_err = ctx->err;
    const hipError_t s0 = hipStreamSynchronize(p.h2d), s1 = hipStreamSynchronize(ctx->stream), s2 = hipStreamSynchronize(p.d2h);
    if (rc != BLUR_OK) { ctx->err = first_err; return rc; }
    for (hipError_t r : { s0, s1, s2 }) step(r);
    if (e != hipSuccess) { ctx->err = std::string("host batch blur: ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    return BLUR_OK;
}

int blur_gaussian_u8c3_host_pitched(blur_ctx* ctx, const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch,
                                    int rows, int cols, double sigma, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    const size_t row_bytes = static_cast<size_t>(cols) * 3;
    if (!src || !dst || rows <= 0 || cols <= 0 || src_pitch < row_bytes || dst_pitch < row_bytes)
        return fail(ctx, BLUR_ERR_INVALID, "null image, non-positive size or pitch shorter than a row");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* dv = nullptr;
    const size_t half = (row_bytes * rows + 255) & ~static_cast<size_t>(255);
    if (int rc0 = ensure_host_stage(ctx, 2 * half, &dv)) return rc0;
    uint8_t* d = static_cast<uint8_t*>(dv);
    int rc = BLUR_OK;
    // the rows are packed on the way in and unpacked on the way out (cv::Mat::step of a ROI or padded Mat); out of place on the device
    hipError_t e = hipMemcpy2DAsync(d, row_bytes, src, src_pitch, row_bytes, rows, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = blur_gaussian_u8c3_dev(ctx, d, d + half, rows, cols, sigma, opts);
    if (e == hipSuccess && rc == BLUR_OK) e = hipMemcpy2DAsync(dst, dst_pitch, d + half, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("host blur (pitched): ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    return rc;
}

int blur_gaussian_f32c1_host(blur_ctx* ctx, const float* src, float* dst, int rows, int cols, double sigma, const blur_opts* opts)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0) return fail(ctx, BLUR_ERR_INVALID, "null plane or non-positive size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(rows) * cols * sizeof(float);
    void* dv = nullptr;
    if (int rc0 = ensure_host_stage(ctx, 2 * bytes, &dv)) return rc0;
    float* d = static_cast<float*>(dv);
    int rc = BLUR_OK;
    hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = blur_gaussian_f32c1_dev(ctx, d, d + static_cast<size_t>(rows) * cols, rows, cols, sigma, opts);
    if (e == hipSuccess && rc == BLUR_OK) e = hipMemcpyAsync(dst, d + static_cast<size_t>(rows) * cols, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("host blur: ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    return rc;
}

// ---- whole-image 2D transform path (pocketfft_2D, Source.cpp:143-277) -----------------------------------------
int blur_pocketfft2d_sizing(int rows, int cols, double sigma, int out[8])
{
    if (rows <= 0 || cols <= 0 || !(sigma > 0) || !out) return BLUR_ERR_INVALID;
    const Sizing2D s = pocketfft2d_sizing(rows, cols, sigma);
    const int v[8] = { s.kSize, s.pad, s.s0, s.s1, s.top, s.bottom, s.left, s.right };
    std::memcpy(out, v, sizeof v);
    return BLUR_OK;
}

int blur_pocketfft2d_u8c3_dev(blur_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int rows, int cols, double sigma, int dft_image, float* d_planes)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_src || !d_dst) return fail(ctx, BLUR_ERR_INVALID, "null image pointer");
    if (rows <= 0 || cols <= 0 || !(sigma > 0)) return fail(ctx, BLUR_ERR_INVALID, "rows, cols and sigma must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const Sizing2D sz = pocketfft2d_sizing(rows, cols, sigma);
    // Reflect_101 clamps a border to dim - 1 (Utils.hpp:217-220) while pocketfft_2D keeps its own sizes[]: past that
    // point the reference's buffers no longer agree with each other
    if (std::max(sz.top, sz.bottom) > rows - 1 || std::max(sz.left, sz.right) > cols - 1)
        return fail(ctx, BLUR_ERR_UNSUPPORTED, "2D path: border > dim - 1 (Reflect_101 would clamp it, Utils.hpp:217-220)");
    if (static_cast<long long>(sz.s0) * sz.s1 > (1ll << 31) - 1) return fail(ctx, BLUR_ERR_UNSUPPORTED, "2D path: ndata exceeds int (Source.cpp:230)");
    DevicePlan *prow = nullptr, *pcol = nullptr;
    if (int rc = get_plan(ctx, sz.s1, false, false, &prow)) return rc;
    if (int rc = get_plan(ctx, sz.s0, false, true, &pcol)) return rc;
    const size_t row_lds = row_lds_bytes(sz.s1);
    if (row_lds > kLdsLimit) return fail(ctx, BLUR_ERR_UNSUPPORTED, "2D path: row FFT length exceeds LDS capacity");
    float *mrow = nullptr, *mcol = nullptr;
    const int *pofy = nullptr, *pofx = nullptr;
    if (dft_image) {
        if (int rc = get_pos_of_freq(ctx, pcol, &pofy)) return rc;
        if (int rc = get_pos_of_freq(ctx, prow, &pofx)) return rc;
    } else {
        // Re(kerf_1D_row[j]) * Re(kerf_1D_col[i]) * (1 / ndata) (:255-263) as two tables in position order, each
        // carrying the 1/n of its own axis; the mirrored upper half of kerf_1D_col (:207) is min(f, n - f) of the table
        if (int rc = get_spectrum(ctx, *prow, sigma, sz.kSize, false, &mrow)) return rc;
        if (int rc = get_spectrum(ctx, *pcol, sigma, sz.kSize, false, &mcol)) return rc;
    }
    if (int rc = ensure_work(ctx, static_cast<size_t>(sz.s0) * sz.s1 * sizeof(float2))) return rc;
    float2* z = reinterpret_cast<float2*>(ctx->work);
    if (int rc = set_lds(ctx, img2d_rows_fwd_kernel, row_lds)) return rc;
    if (int rc = set_lds(ctx, img2d_rows_inv_kernel, row_lds)) return rc;
    const Img2dGeom g{ rows, cols, sz.s0, sz.s1, sz.top, sz.left };
    ctx->last_family = 3;
    const int pairs[2][2] = { { 0, 1 }, { 2, -1 } };
    for (const auto& pr : pairs) {
        hipLaunchKernelGGL(img2d_rows_fwd_kernel, dim3(sz.s0), dim3(kThreads), row_lds, ctx->stream, d_src, z, g, pr[0], pr[1], prow->dev, prow->d_tw);
        HIP_TRY(ctx, hipGetLastError());
        if (dft_image) {
            if (int rc = launch_img2d_cols<false>(ctx, z, sz.s0, sz.s1, *pcol, nullptr, nullptr)) return rc;
            const size_t px = static_cast<size_t>(rows) * cols;
            const int grid = static_cast<int>(std::min<size_t>((px + 255) / 256, static_cast<size_t>(ctx->num_cus) * 16));
            hipLaunchKernelGGL(img2d_logspec_kernel, dim3(grid), dim3(256), 0, ctx->stream, z, d_dst, d_planes, g, pr[0], pr[1], pofy, pofx);
        } else {
            if (int rc = launch_img2d_cols<true>(ctx, z, sz.s0, sz.s1, *pcol, mcol, mrow)) return rc;
            hipLaunchKernelGGL(img2d_rows_inv_kernel, dim3(rows), dim3(kThreads), row_lds, ctx->stream, z, d_dst, d_planes, g, pr[0], pr[1], prow->dev, prow->d_tw);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    return BLUR_OK;
}

int blur_pocketfft2d_u8c3_host(blur_ctx* ctx, const uint8_t* src, uint8_t* dst, int rows, int cols, double sigma, int dft_image)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0) return fail(ctx, BLUR_ERR_INVALID, "null image or non-positive size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(rows) * cols * 3;
    void* dv = nullptr;
    if (int rc0 = ensure_host_stage(ctx, bytes, &dv)) return rc0;
    uint8_t* d = static_cast<uint8_t*>(dv);
    int rc = BLUR_OK;
    hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    // in place on the device, like the reference (image.data is source and destination): the pass over planes 0 and 1
    // writes only their bytes, and the later pass over plane 2 reads only plane 2
    if (e == hipSuccess) rc = blur_pocketfft2d_u8c3_dev(ctx, d, d, rows, cols, sigma, dft_image, nullptr);
    if (e == hipSuccess && rc == BLUR_OK) e = hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("host 2D path: ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    return rc;
}

int blur_reflect101_u8_dev(blur_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, int rows, int cols, int channels, int top, int bottom, int left, int right,
                           int out_size[2])
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (rows <= 0 || cols <= 0 || channels <= 0 || top < 0 || bottom < 0 || left < 0 || right < 0)
        return fail(ctx, BLUR_ERR_INVALID, "Reflect_101: bad arguments");
    // the reference's only difference to cv::copyMakeBorder: borders are clamped to dim - 1 (Utils.hpp:217-220)
    top = std::min(top, rows - 1);
    bottom = std::min(bottom, rows - 1);
    left = std::min(left, cols - 1);
    right = std::min(right, cols - 1);
    const int orows = rows + top + bottom, ocols = cols + left + right;
    if (out_size) { out_size[0] = orows; out_size[1] = ocols; }
    if (!d_out) return BLUR_OK;                                  // size query
    if (!d_in || d_in == d_out) return fail(ctx, BLUR_ERR_INVALID, "Reflect_101: null input or in place");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = static_cast<size_t>(orows) * ocols;
    const int grid = static_cast<int>(std::min<size_t>((px + 255) / 256, static_cast<size_t>(ctx->num_cus) * 16));
    hipLaunchKernelGGL(reflect101_u8_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_in, d_out, rows, cols, channels, top, left, orows, ocols);
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

int blur_flip_block_f32_dev(blur_ctx* ctx, const float* d_in, float* d_out, int w, int h)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_in || !d_out || w <= 0 || h <= 0 || d_in == d_out) return fail(ctx, BLUR_ERR_INVALID, "flip_block: bad arguments (out of place only)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int tiles = ((w + 63) / 64) * ((h + 63) / 64);
    hipLaunchKernelGGL(flip_block_kernel, dim3(tiles), dim3(256), 0, ctx->stream, d_in, d_out, w, h);
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

int blur_deinterleave_bgr_u8_f32_dev(blur_ctx* ctx, const uint8_t* d_in, float* d_planes, uint32_t total)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_in || !d_planes) return fail(ctx, BLUR_ERR_INVALID, "null pointer");
    if (!total) return BLUR_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const unsigned grid = std::min<uint32_t>((total + 255) / 256, 2048u * 8);
    hipLaunchKernelGGL(deinterleave_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_in, d_planes, total);
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

int blur_interleave_bgr_f32_u8_dev(blur_ctx* ctx, const float* d_planes, uint8_t* d_out, uint32_t total)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_planes || !d_out) return fail(ctx, BLUR_ERR_INVALID, "null pointer");
    if (!total) return BLUR_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const unsigned grid = std::min<uint32_t>((total + 255) / 256, 2048u * 8);
    hipLaunchKernelGGL(interleave_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_planes, d_out, total);
    HIP_TRY(ctx, hipGetLastError());
    return BLUR_OK;
}

// BLUR_BOX_NO_MFMA=1 (developer switch): the accumulator kernels only
static bool box_no_mfma()
{
    static const bool off = [] { const char* e = getenv("BLUR_BOX_NO_MFMA"); return e && *e && *e != '0'; }();
    return off;
}

int blur_fastboxblur_u8_dev(blur_ctx* ctx, uint8_t* d_inout, int w, int h, int channels, int ksize, int passes)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_inout || w <= 0 || h <= 0 || channels <= 0 || ksize <= 0 || passes < 0)
        return fail(ctx, BLUR_ERR_INVALID, "fastboxblur: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(w) * h * channels;
    // the second image, and behind it the mirrored row margins of the matrix-core horizontal kernel (bx_box.hip)
    const size_t margins_at = (bytes + 255) & ~static_cast<size_t>(255);
    const size_t margins_bytes = bx_horizontal_scratch(h, w, channels, std::min((ksize - 1) / 2, w - 1), std::min(passes, 3));
    if (ctx->box_bytes < margins_at + margins_bytes) {
        if (ctx->box_tmp) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(ctx->box_tmp)); ctx->box_tmp = nullptr; ctx->box_bytes = 0; }
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->box_tmp), margins_at + margins_bytes));
        ctx->box_bytes = margins_at + margins_bytes;
    }
    uint8_t *a = d_inout, *b = ctx->box_tmp;
    uint8_t* margins = margins_bytes ? ctx->box_tmp + margins_at : nullptr;
    auto sweep = [&](int nlines, int n, size_t lstride, size_t xstride) {
        int r = (ksize - 1) / 2;
        if (r > n - 1) r = n - 1;
        const int seg_len = std::max(64, (n + 31) / 32);
        const int nseg = (n + seg_len - 1) / seg_len;
        const long long items = static_cast<long long>(nlines) * channels * nseg;
        const unsigned grid = static_cast<unsigned>((items + 255) / 256);
        hipLaunchKernelGGL(boxsweep_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, b, nlines, n, lstride, xstride, channels, r, seg_len);
        std::swap(a, b);
    };
    // horizontal sweeps: all passes of a row inside LDS when two copies of the row fit
    const size_t row_lds = 2 * ((static_cast<size_t>(w) * channels + 3) & ~static_cast<size_t>(3));
    // four-pixel-per-step kernel: row + physical reflect borders (r + 1 pixels) + slack for whole groups, twice
    int r_row = (ksize - 1) / 2;
    if (r_row > w - 1) r_row = w - 1;
    const int off0 = ((r_row + 2) * channels + 4 + 15) & ~15;                       // whole aligned dwords left of pixel -(r+1)
    const int buf_bytes = (off0 + (w + r_row + 8) * channels + 16 + 15) & ~15;       // ... and right of pixel w+r+3
    // horizontal sweeps: up to three at a time in one launch on the integer matrix cores where that kernel applies
    int hdone = 0;
    if (!box_no_mfma() && r_row > 0) {
        while (hdone < passes) {
            const int now = std::min(3, passes - hdone);
            bool ran = false;
            HIP_TRY(ctx, bx_horizontal(ctx->stream, a, b, margins, h, w, channels, r_row, now, ctx->num_cus, &ran));
            if (!ran) break;
            std::swap(a, b);
            hdone += now;
        }
    }
    if (r_row == 0) hdone = passes;               // a box of one pixel
    const int hleft = passes - hdone;
    if (hleft == 0) {
    } else if ((channels == 1 || channels == 3 || channels == 4) && 2 * static_cast<size_t>(buf_bytes) <= kLdsLimit) {
        const int gpt = ((w + 3) / 4 + 255) / 256;
        const size_t lds = 2 * static_cast<size_t>(buf_bytes);
        auto launch = [&](auto kern) -> int {
            if (int rc = set_lds(ctx, kern, lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(h), dim3(256), lds, ctx->stream, a, b, w, r_row, hleft, gpt, off0, buf_bytes);
            return BLUR_OK;
        };
        int rc = channels == 1 ? launch(boxrow4_kernel<1>) : channels == 3 ? launch(boxrow4_kernel<3>) : launch(boxrow4_kernel<4>);
        if (rc) return rc;
        std::swap(a, b);
    } else if (row_lds <= kLdsLimit) {
        if (int rc = set_lds(ctx, boxrow_kernel, row_lds)) return rc;
        int r = (ksize - 1) / 2;
        if (r > w - 1) r = w - 1;
        const int nseg = std::max(1, 256 / channels);
        const int seg_len = (w + nseg - 1) / nseg;
        hipLaunchKernelGGL(boxrow_kernel, dim3(h), dim3(256), row_lds, ctx->stream, a, b, w, channels, r, hleft, seg_len, nseg);
        std::swap(a, b);
    } else {
        for (int p = 0; p < hleft; ++p) sweep(h, w, static_cast<size_t>(w) * channels, static_cast<size_t>(channels));
    }
    // vertical sweeps: up to three at a time in one launch on the integer matrix cores (bx_box.hip) where that kernel applies
    int vdone = 0;
    if (!box_no_mfma()) {
        int r = (ksize - 1) / 2;
        if (r > h - 1) r = h - 1;
        while (vdone < passes && r > 0) {
            const int now = std::min(3, passes - vdone);
            bool ran = false;
            HIP_TRY(ctx, bx_vertical(ctx->stream, a, b, h, static_cast<int>(static_cast<size_t>(w) * channels), r, now, ctx->num_cus, &ran));
            if (!ran) break;
            std::swap(a, b);
            vdone += now;
        }
        if (r == 0) vdone = passes;            // a box of one row
    }
    if (vdone == passes) {
    } else if ((static_cast<size_t>(w) * channels) % 4 == 0) {
        const int pitch4 = static_cast<int>(static_cast<size_t>(w) * channels / 4);
        int r = (ksize - 1) / 2;
        if (r > h - 1) r = h - 1;
        // enough (dword column, segment) items to fill the chip a few times over
        int nseg = std::max(1, std::min(h / std::max(2 * r + 1, 64), (256 * 2048) / std::max(1, pitch4) + 1));
        const int seg_len = (h + nseg - 1) / nseg;
        nseg = (h + seg_len - 1) / seg_len;
        const long long items = static_cast<long long>(pitch4) * nseg;
        for (int p = vdone; p < passes; ++p) {
            hipLaunchKernelGGL(boxcol4_kernel, dim3(static_cast<unsigned>((items + 255) / 256)), dim3(256), 0, ctx->stream, a, b, h, pitch4, r, seg_len, nseg);
            std::swap(a, b);
        }
    } else {
        for (int p = vdone; p < passes; ++p) sweep(w, h, static_cast<size_t>(channels), static_cast<size_t>(w) * channels);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (a != d_inout) HIP_TRY(ctx, hipMemcpyAsync(d_inout, a, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return BLUR_OK;
}

int blur_fastboxblur_u8_host(blur_ctx* ctx, uint8_t* inout, int w, int h, int channels, int ksize, int passes)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!inout || w <= 0 || h <= 0 || channels <= 0) return fail(ctx, BLUR_ERR_INVALID, "fastboxblur: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(w) * h * channels;
    void* dv = nullptr;
    if (int rc0 = ensure_host_stage(ctx, bytes, &dv)) return rc0;
    uint8_t* d = static_cast<uint8_t*>(dv);
    int rc = BLUR_OK;
    hipError_t e = hipMemcpyAsync(d, inout, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = blur_fastboxblur_u8_dev(ctx, d, w, h, channels, ksize, passes);
    if (e == hipSuccess && rc == BLUR_OK) e = hipMemcpyAsync(inout, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("host fastboxblur: ") + hipGetErrorString(e); return BLUR_ERR_HIP; }
    return rc;
}

/* diagnostic builds (-DFK_STAMPS): copy the stamp tail behind the multiplier table of FFT length n
   (role 1 = specialised row plan, 2 = specialised column plan) */
// ---- several GPUs from one host thread ------------------------------------------------------------
// Frames are independent (Source.cpp:510 even runs channels serially), so a batch shards by frame with no exchange
// between the shards: shard r gets frames [n r / S, n (r + 1) / S) and its own context and stream.  `devices` may
// repeat an ordinal: several logical shards on one GPU (that is also how the path is tested on a one-GPU box).
struct blur_multi {
    std::vector<int> devices;
    std::vector<blur_ctx*> ctxs;
    std::vector<hipStream_t> streams;
    std::vector<uint8_t*> stage;        // per shard: frames of this shard on its device (shards away from the frames' device)
    std::vector<size_t> stage_bytes;
    std::string err;
};

int blur_multi_create(blur_multi** out, const int* devices, int ndevices)
{
    if (!out || !devices || ndevices <= 0) return BLUR_ERR_INVALID;
    *out = nullptr;
    auto m = std::make_unique<blur_multi>();
    for (int r = 0; r < ndevices; ++r) {
        blur_ctx* c = nullptr;
        int rc = blur_ctx_create(&c, devices[r]);
        hipStream_t st = nullptr;
        if (rc == BLUR_OK && (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)) rc = BLUR_ERR_HIP;
        if (rc == BLUR_OK) rc = blur_ctx_set_stream(c, st);
        if (rc != BLUR_OK) {
            if (c) blur_ctx_destroy(c);
            if (st) (void)hipStreamDestroy(st);
            for (size_t i = 0; i < m->ctxs.size(); ++i) { blur_ctx_destroy(m->ctxs[i]); (void)hipStreamDestroy(m->streams[i]); }
            return rc;
        }
        m->devices.push_back(devices[r]);
        m->ctxs.push_back(c);
        m->streams.push_back(st);
        m->stage.push_back(nullptr);
        m->stage_bytes.push_back(0);
    }
    *out = m.release();
    return BLUR_OK;
}

int blur_multi_destroy(blur_multi* m)
{
    if (!m) return BLUR_ERR_INVALID;
    for (size_t r = 0; r < m->ctxs.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        (void)hipStreamSynchronize(m->streams[r]);
        if (m->stage[r]) (void)hipFree(m->stage[r]);
        blur_ctx_destroy(m->ctxs[r]);          // synchronises its stream before freeing
        (void)hipStreamDestroy(m->streams[r]);
    }
    delete m;
    return BLUR_OK;
}

int blur_multi_shards(const blur_multi* m) { return m ? static_cast<int>(m->ctxs.size()) : 0; }
const char* blur_multi_last_error(const blur_multi* m) { return m ? m->err.c_str() : "null blur_multi"; }

// location: 0 = src/dst are host pointers (pinned memory overlaps the copies of different shards), 1 = device pointers
// on devices[0].  Synchronous: returns when every shard is done.
static int blur_multi_run(blur_multi* m, const uint8_t* src, uint8_t* dst, int nframes, int rows, int cols, double sigma,
                          const blur_opts* opts, int location)
{
    if (!m) return BLUR_ERR_INVALID;
    if (!src || !dst || nframes < 0 || rows <= 0 || cols <= 0) { m->err = "null frame pointer, negative frame count or non-positive size"; return BLUR_ERR_INVALID; }
    const int S = static_cast<int>(m->ctxs.size());
    const size_t fb = static_cast<size_t>(rows) * cols * 3;
    int rc_all = BLUR_OK;
    for (int r = 0; r < S && rc_all == BLUR_OK; ++r) {
        const int b = static_cast<int>(static_cast<long long>(nframes) * r / S), e = static_cast<int>(static_cast<long long>(nframes) * (r + 1) / S);
        if (e <= b) continue;
        const size_t bytes = fb * (e - b);
        blur_ctx* c = m->ctxs[r];
        hipStream_t st = m->streams[r];
        auto hip_fail = [&](hipError_t er, const char* what) { m->err = std::string(what) + ": " + hipGetErrorString(er); rc_all = BLUR_ERR_HIP; };
        if (hipError_t er = hipSetDevice(m->devices[r]); er != hipSuccess) { hip_fail(er, "hipSetDevice"); break; }
        const bool local = location == 1 && m->devices[r] == m->devices[0];
        const uint8_t* in = src + fb * b;
        uint8_t* outp = dst + fb * b;
        uint8_t* work_in = const_cast<uint8_t*>(in);
        uint8_t* work_out = outp;
        if (!local) {
            if (m->stage_bytes[r] < bytes) {
                if (m->stage[r]) { (void)hipStreamSynchronize(st); (void)hipFree(m->stage[r]); m->stage[r] = nullptr; m->stage_bytes[r] = 0; }
                if (hipError_t er = hipMalloc(reinterpret_cast<void**>(&m->stage[r]), bytes); er != hipSuccess) { hip_fail(er, "hipMalloc (shard staging)"); break; }
                m->stage_bytes[r] = bytes;
            }
            work_in = work_out = m->stage[r];
            const hipError_t er = location == 0 ? hipMemcpyAsync(work_in, in, bytes, hipMemcpyHostToDevice, st)
                                                : hipMemcpyPeerAsync(work_in, m->devices[r], in, m->devices[0], bytes, st);
            if (er != hipSuccess) { hip_fail(er, "fan-out copy"); break; }
        }
        const int rc = blur_gaussian_u8c3_batch_dev(c, work_in, work_out, e - b, rows, cols, sigma, opts);
        if (rc != BLUR_OK) { m->err = std::string("shard ") + std::to_string(r) + ": " + blur_last_error(c); rc_all = rc; break; }
        if (!local) {
            const hipError_t er = location == 0 ? hipMemcpyAsync(outp, work_out, bytes, hipMemcpyDeviceToHost, st)
                                                : hipMemcpyPeerAsync(outp, m->devices[0], work_out, m->devices[r], bytes, st);
            if (er != hipSuccess) { hip_fail(er, "fan-in copy"); break; }
        }
    }
    for (int r = 0; r < S; ++r) {
        (void)hipSetDevice(m->devices[r]);
        const hipError_t er = hipStreamSynchronize(m->streams[r]);
        if (er != hipSuccess && rc_all == BLUR_OK) { m->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(er); rc_all = BLUR_ERR_HIP; }
    }
    return rc_all;
}

int blur_gaussian_u8c3_batch_multi_dev(blur_multi* m, const uint8_t* d_src, uint8_t* d_dst, int nframes, int rows, int cols, double sigma, const blur_opts* opts)
{
    if (m) {
        // frames queued by the caller on devices[0] must be complete before other devices (and other streams) read them
        if (hipSetDevice(m->devices[0]) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { m->err = "hipDeviceSynchronize on the frames' device failed"; return BLUR_ERR_HIP; }
    }
    return blur_multi_run(m, d_src, d_dst, nframes, rows, cols, sigma, opts, 1);
}

int blur_gaussian_u8c3_batch_multi_host(blur_multi* m, const uint8_t* src, uint8_t* dst, int nframes, int rows, int cols, double sigma, const blur_opts* opts)
{
    return blur_multi_run(m, src, dst, nframes, rows, cols, sigma, opts, 0);
}

int blur_convolve_lines_c32_dev(blur_ctx* ctx, const float* d_in, float* d_out, int nlines, int n, const float* multipliers)
{
    if (!ctx) return BLUR_ERR_INVALID;
    if (!d_in || !d_out || !multipliers || nlines < 0 || n <= 0) return fail(ctx, BLUR_ERR_INVALID, "null pointer or non-positive size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const WrEntry* e = nullptr;
    for (int role = 0; role < 2 && !e; ++role) {
        const WrEntry* c = find_wr_entry(n, role != 0);
        if (c && c->r0 * kWrS == n) e = c;
    }
    if (!e) return fail(ctx, BLUR_ERR_UNSUPPORTED, "no wave-resident kernel for this length (blur_wr_length)");
    float2* tw0 = nullptr;
    if (int rc = wr_get_tables(ctx, e, &tw0)) return rc;
    // the caller's multiplier tables, cached by content (hash AND a full comparison: a colliding table would give a wrong image
    // silently); at most kLinesCacheMax tables are kept, the least recently used one goes first
    constexpr size_t kLinesCacheMax = 16;
    float* d_m = nullptr;
    const uint64_t h = fnv1a(multipliers, sizeof(float) * n);
    for (auto it = ctx->lines_tables.begin(); it != ctx->lines_tables.end(); ++it) {
        if (it->hash == h && static_cast<int>(it->host.size()) == n && std::memcmp(it->host.data(), multipliers, sizeof(float) * n) == 0) {
            d_m = it->dev;
            ctx->lines_tables.splice(ctx->lines_tables.begin(), ctx->lines_tables, it);      // most recently used first
            break;
        }
    }
    if (!d_m) {
        if (ctx->lines_tables.size() >= kLinesCacheMax) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                                   // a launch may still read the table
            (void)hipFree(ctx->lines_tables.back().dev);
            ctx->lines_tables.pop_back();
        }
        // (the tail behind the table is only written by -DWR_STAMPS diagnostic builds, as in wr_get_spectrum)
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_m), sizeof(float) * (n + kWrStampTailFloats)));
        if (hipMemset(d_m, 0, sizeof(float) * (n + kWrStampTailFloats)) != hipSuccess ||
            hipMemcpy(d_m, multipliers, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_m);
            return fail(ctx, BLUR_ERR_HIP, "blur_convolve_lines_c32_dev: table upload failed");
        }
        ctx->lines_tables.push_front({ h, std::vector<float>(multipliers, multipliers + n), d_m });
    }
    HIP_TRY(ctx, e->lines(ctx->stream, reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out), nlines, ctx->num_cus, ctx->d_w256, tw0, d_m));
    return BLUR_OK;
}

int blur_mx_fragments(const float* taps, int pad, int nkb, uint16_t* out)
{
    if (!taps || !out || pad < 0 || nkb < mx_nkb(pad)) return BLUR_ERR_INVALID;
    mx_fragments(taps, pad, nkb, out);
    return BLUR_OK;
}

int blur_mx_window_blocks(int pad)
{
    const MxEntry* e = pad >= 0 ? find_mx_entry(pad) : nullptr;
    return e ? e->nkb : 0;
}

int blur_wr_length(int need, int column_role)
{
    const WrEntry* e = find_wr_entry(need, column_role != 0);
    return e ? e->r0 * kWrS : 0;
}

// which kernels the last 8-bit 3-channel blur of this context ran: 0 run-time-planned, 1 specialised rows-first (both passes),
// 2 wave-resident, 3 whole-image 2D transform, 4 two-kernel matrix-core engine, 6 fused matrix-core kernel; -1 none yet
// (bench.py --preset reference-sweep reports it per size)
int blur_debug_last_family(const blur_ctx* ctx) { return ctx ? ctx->last_family : -1; }

int blur_last_engine(const blur_ctx* ctx, char* note, size_t n)
{
    if (!ctx) return -1;
    static const char* const names[] = { "run-time-planned FFT kernels", "specialised rows-first FFT kernels", "wave-resident FFT kernels", "whole-image 2D FFT",
                                         "two-kernel matrix-core engine", "?", "fused matrix-core kernel", "tiled wave-resident FFT kernels" };
    if (note && n > 0) {
        std::string t = ctx->last_family >= 0 && ctx->last_family <= 7 ? names[ctx->last_family] : "none yet";
        if (!ctx->engine_note.empty()) t += " (not taken: " + ctx->engine_note + ")";
        std::snprintf(note, n, "%s", t.c_str());
    }
    return ctx->last_family;
}

// redzone tests: number of bytes of the workspace's two guard bands that were overwritten (0 = intact; -1 = no workspace yet)
int blur_debug_check_workspace_guards(blur_ctx* ctx)
{
    if (!ctx) return -1;
    if (!ctx->work) return -1;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
    std::vector<unsigned char> g(2 * kWorkGuard);
    const char* base = reinterpret_cast<const char*>(ctx->work) - kWorkGuard;
    if (hipMemcpy(g.data(), base, kWorkGuard, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(g.data() + kWorkGuard, base + kWorkGuard + ctx->work_bytes, kWorkGuard, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int bad = 0;
    for (unsigned char c : g) bad += c != 0xA5;
    return bad;
}

int blur_debug_read_stamps(blur_ctx* ctx, int n, int role, unsigned long long* out, int count)
{
    if (!ctx || !out) return BLUR_ERR_INVALID;
    auto it = ctx->last_spectrum.find(role < 0 ? -n : 4 * n + role);      // role < 0: the wave-resident kernel of length n
    if (it == ctx->last_spectrum.end() || count * sizeof(unsigned long long) > kStampTailFloats * sizeof(float)) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, it->second + n, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return BLUR_OK;
}

int blur_malloc(blur_ctx* ctx, void** d_ptr, size_t bytes)
{
    if (!ctx || !d_ptr) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return BLUR_OK;
}

int blur_host_alloc(blur_ctx* ctx, void** h_ptr, size_t bytes)
{
    if (!ctx || !h_ptr) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return BLUR_OK;
}

int blur_host_free(blur_ctx* ctx, void* h_ptr)
{
    if (!ctx) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipHostFree(h_ptr));
    return BLUR_OK;
}

int blur_free(blur_ctx* ctx, void* d_ptr)
{
    if (!ctx) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipFree(d_ptr));
    return BLUR_OK;
}

int blur_memcpy_h2d(blur_ctx* ctx, void* d_dst, const void* src, size_t bytes)
{
    if (!ctx) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BLUR_OK;
}

int blur_memcpy_d2h(blur_ctx* ctx, void* dst, const void* d_src, size_t bytes)
{
    if (!ctx) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BLUR_OK;
}

int blur_copy_bandwidth(blur_ctx* ctx, size_t bytes, int reps, double* gbs)
{
    if (!ctx || !gbs || reps <= 0 || bytes < 4096) return BLUR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n16 = bytes / 16;
    uint4 *a = nullptr, *b = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&a), n16 * 16));
    if (hipMalloc(reinterpret_cast<void**>(&b), n16 * 16) != hipSuccess) { (void)hipFree(a); return fail(ctx, BLUR_ERR_NOMEM, "blur_copy_bandwidth: out of device memory"); }
    (void)hipMemsetAsync(a, 0x5a, n16 * 16, ctx->stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = ctx->num_cus * 8;
    hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, b, n16);
    (void)hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, b, n16);
    (void)hipEventRecord(e1, ctx->stream);
    hipError_t err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (err != hipSuccess || !(ms > 0.f)) return fail(ctx, BLUR_ERR_HIP, "blur_copy_bandwidth: timing failed");
    *gbs = 2.0 * static_cast<double>(n16) * 16.0 * reps / (ms * 1e-3) / 1e9;
    return BLUR_OK;
}

}  // extern "C"
