// fastboxblur (call site Source.cpp:587; semantics: oracle/boxblur_oracle.c) on v_mfma_i32_16x16x64_i8.
//
// One sweep of a box of 2 r + 1 taps is a banded matrix of ones applied along the sweep direction.  A wave walks along that
// direction in steps of 16 positions and keeps, per 16 lines and per sweep, a WINDOW of 64 NB positions as the B operand of the
// matrix instruction (lane = line, 16 bytes = 16 positions each of the 4 lane groups: exactly 64 per instruction); the band is the
// A operand (constants), so ONE instruction (NB of them for boxes wider than 49) gives the 16 x 16 running sums of the step -- no
// recurrence, nothing to set up when a segment starts.  The accumulator comes out with the line on the lane and four consecutive
// positions in its four registers, which is, byte for byte, the layout of the next window dword of the SAME lane: the u8 result of
// sweep p (one v_mul_hi_u32_u24 per byte, below) is the input of sweep p + 1 with no lane movement, and the P sweeps run as a
// pipeline inside the wave: image bytes are read once and written once per direction (6 B/px for three channels instead of 6 P).
//
// Signed operands.  The instruction multiplies signed bytes, so windows hold x - 128 (x ^ 0x80) and the sums are put right by the
// accumulator's start value.
//
// Rounding.  The oracle's (uint8)(int)(acc * (1.f / n) + 0.5f), n = 2 r + 1 odd, equals floor((acc + r) / n): acc / n + 1/2 is
// never closer than 1 / (2 n) to an integer and the float error is below 5e-5.  The band's entries are s (64; 127 for n = 3) and
// the accumulator starts at s r, so the instruction delivers s (acc + r) < 2^24, and the high half of its 24 x 24-bit product with
// M = ceil(2^32 / (s n)) is that floor exactly (error term (acc + r + 256 n) s n / 2^32 < 1 / n for n < 400) -- written straight into
// byte k of the packed dword by the SDWA form of the multiply.
//
// Borders.  The blur of a reflect-101 extension is the extension of the blur (the window is symmetric and the sums are integers),
// so margins are never special: the vertical kernel reads mirrored rows, the horizontal one reads mirrored margins that a small
// kernel lays out beside the image first; the sweeps' intermediate values at positions outside the image are then the mirrored
// intermediate values, as the oracle's per-sweep reflect requires.
#include "bx_box.hpp"
#include <algorithm>
#include <cstdlib>

namespace blur_amd {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kRsrcWord3 = 0x00020000u;   // raw buffer, 32-bit data format (gfx9 family)
constexpr uint32_t kDropped = 0xfffffff0u;     // an offset past every buffer: the store is dropped, the load returns 0

template <int CTRL> __device__ __forceinline__ uint32_t bx_dpp(uint32_t x)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xf, 0xf, true));
}

// 4 x 4 byte transpose inside a lane quad: lane j holds bytes M[j][0..3]; afterwards lane k holds M[0..3][k]
__device__ __forceinline__ uint32_t bx_quad_transpose(uint32_t p, uint32_t sel1, uint32_t sel2)
{
    const uint32_t t = bx_dpp<0xb1>(p);                       // quad_perm [1,0,3,2]
    const uint32_t q = __builtin_amdgcn_perm(t, p, sel1);
    const uint32_t u = bx_dpp<0x4e>(q);                       // quad_perm [2,3,0,1]
    return __builtin_amdgcn_perm(u, q, sel2);
}

// the four sums of a lane -> four bytes of one dword: floor(sum / (s n)) each (sum = s (acc + r), see the header); the low byte
// of each quotient (the sweeps that feed another sweep deliver v + 128 < 384: the byte wraps to v ^ 0x80).  Plain C on purpose:
// the compiler pads the matrix-result -> vector-read and vector-write -> matrix-operand hazards of its own instructions only
// (an SDWA form of the multiply, written as inline assembly, packs in 4 instructions instead of 7 but its last byte reached the
// next matrix instruction too late: +-1 on 2 % of the bytes for r >= 12).
__device__ __forceinline__ uint32_t bx_mulhi24(int d, uint32_t mul)
{
    return static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(d) & 0xffffffu) * (mul & 0xffffffu)) >> 32);
}
__device__ __forceinline__ uint32_t bx_pack(v4i d, uint32_t mul)
{
    const uint32_t lo = __builtin_amdgcn_perm(bx_mulhi24(d[1], mul), bx_mulhi24(d[0], mul), 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm(bx_mulhi24(d[3], mul), bx_mulhi24(d[2], mul), 0x04000c0cu);
    return lo | hi;
}

// The band.  The window of a sweep is a ring of W = 4 NB dwords per lane (newest at ring slot step % W); dword d of operand group
// G is ring slot 4 G + d and holds, in lane group q, the positions 16 t + 4 q + i (i = byte) of window tile t = W - 1 - age,
// age = (step - slot) mod W.  The step's 16 outputs are the window's positions DELTA .. DELTA + 15, DELTA = 32 NB - 8 (the
// middle), output m on lane m of the A operand.  Fragment u = (step - 4 G) mod W serves group G: bx_band(u)[d] byte i is s where
// |16 t + 4 q + i - DELTA - m| <= reach with age = (u - d) mod W.  `stride`: taps sit on every stride-th position (the channel
// count for the horizontal sweeps over interleaved pixels, 1 for the vertical ones); reach = stride r.
template <int NB> __device__ __forceinline__ v4i bx_band(int u, int m, int q, int r, int stride, int s)
{
    constexpr int W = 4 * NB, DELTA = 32 * NB - 8;
    const int dm = stride == 3 ? 171 : 1, ds = stride == 3 ? 9 : stride == 4 ? 2 : 0;
    v4i a;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int age = (u - d + W) % W, t = W - 1 - age;
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = 16 * t + 4 * q + i - DELTA - m, ao = off < 0 ? -off : off;
            const int rem = ao - ((ao * dm) >> ds) * stride;      // ao % stride without a division (stride 1, 3 or 4; ao < 256)
            word |= (ao <= stride * r && rem == 0 ? static_cast<uint32_t>(s) : 0u) << (8 * i);
        }
        a[d] = static_cast<int>(word);
    }
    return a;
}

// ---- vertical ------------------------------------------------------------------------------------------------------------
// A wave owns a strip of 16 NT byte columns and a segment of rows.  Lane (n = l & 15, q = l >> 4), quad qd = n >> 2, p = n & 3:
// per step it loads NT dwords (as dwordx4) of row 16 step + 4 q + p, bytes 4 NT qd .. of the strip -- a wave's load covers 16 rows
// x 16 NT contiguous bytes -- and a 4 x 4 byte transpose inside the quad turns dword t into "column 16 qd + 4 t + p of the strip,
// rows 4 q .. 4 q + 3": the window dword of tile t (tile t = the strip's columns 16 qd' + 4 t + p', any assignment of columns to
// tiles will do).  Stores take the same road back.  Sweep p's tile of step s covers rows R0 - p DELTA + 16 s ..: the walk starts
// P DELTA rows above the segment and the first P (W - 1) steps only fill the pipeline.  (Skewing the sweeps by one step each, so
// that the P products of an iteration are independent of each other, changed nothing in the horizontal kernel and cost 12 % in
// this one: the chain is hidden by the other waves already.)
template <int NB, int P, int NT>
__global__ __launch_bounds__(256) void bx_vert_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int h, int pitch, int r, int s_band, uint32_t mul,
                                                      int seg_rows, int nstrips, int nwaves)
{
    constexpr int W = 4 * NB, DELTA = 32 * NB - 8, FILL = P * (W - 1);
    static_assert(NT % 4 == 0, "dwordx4 loads");
    const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * 4 + (threadIdx.x >> 6)));      // wave-uniform: scalar branches below
    if (wid >= nwaves) return;
    const int l = threadIdx.x & 63, n = l & 15, q = l >> 4, qd = n >> 2, p = n & 3;
    // (Walking every other segment of a strip upwards, so that two neighbours read the P DELTA rows either side of their boundary at
    // the same time, with or without the four segments of a strip in one workgroup, was measured at 8K: 53.5 and 69 us against 52 --
    // the re-read rows do not cost memory bandwidth, and four adjacent strips per workgroup, 256 contiguous bytes of a row, matter.)
    const int strip = wid % nstrips, seg = wid / nstrips;
    const int ys = seg * seg_rows, ye = min(h, ys + seg_rows);
    const int S = FILL + (ye - ys + 15) / 16, R0 = ys - P * DELTA;
    const uint32_t col0 = static_cast<uint32_t>(strip) * (16 * NT) + 4 * NT * qd;
    const uint32_t sel1 = (l & 1) ? 0x03070105u : 0x06020400u, sel2 = (l & 2) ? 0x03020706u : 0x05040100u;
    const uint32_t bytes = static_cast<uint32_t>(h) * static_cast<uint32_t>(pitch);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in), 0, bytes, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, kRsrcWord3);
    const bool whole16 = (pitch & 15) == 0;

    v4i band[W];
#pragma unroll
    for (int u = 0; u < W; ++u) band[u] = bx_band<NB>(u, n, q, r, 1, s_band);
    // the instruction's operands are SIGNED bytes: windows hold x ^ 0x80 = x - 128, the accumulator starts at s (128 n + r) more,
    // and the sweeps that feed another sweep add 128 to their result on the way out (another s 128 n: the byte wraps to v ^ 0x80)
    const int nt = 2 * r + 1, bias_last = s_band * (128 * nt + r), bias_mid = s_band * (256 * nt + r);
    const v4i cin_last = { bias_last, bias_last, bias_last, bias_last }, cin_mid = { bias_mid, bias_mid, bias_mid, bias_mid };

    uint32_t win[P][NT][W];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int d = 0; d < W; ++d) win[a][t][d] = 0;

#ifndef BX_VPF
#define BX_VPF 2
#endif
    constexpr int VPF = BX_VPF;              // steps of loads in flight
    static_assert((2 * W) % VPF == 0, "the prefetch ring is indexed statically");
    u4 ld[VPF][NT / 4];
    auto issue = [&](int s, u4 (&dst)[NT / 4]) __attribute__((always_inline)) {
        int row = R0 + 16 * s + 4 * q + p;
        row = row < 0 ? -row : row;
        row = row >= h ? 2 * (h - 1) - row : row;
        const uint32_t off = static_cast<uint32_t>(row) * static_cast<uint32_t>(pitch) + col0;
#pragma unroll
        for (int k = 0; k < NT / 4; ++k) dst[k] = __builtin_amdgcn_raw_buffer_load_b128(rin, off + 16 * k, 0, 0);
    };
#pragma unroll
    for (int k = 0; k < VPF; ++k)
        if (k < S) issue(k, ld[k]);

    for (int s0 = 0; s0 < S; s0 += 2 * W) {
#pragma unroll
        for (int uu = 0; uu < 2 * W; ++uu) {
            const int s = s0 + uu, slot = uu % W;
            if (s >= S) break;
            u4 (&cur)[NT / 4] = ld[uu % VPF];
            uint32_t res[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                win[0][t][slot] = bx_quad_transpose(cur[t / 4][t % 4], sel1, sel2) ^ 0x80808080u;
#pragma unroll
                for (int a = 0; a < P; ++a) {
                    v4i d = a + 1 < P ? cin_mid : cin_last;
#pragma unroll
                    for (int G = 0; G < NB; ++G) {
                        const v4i b = { static_cast<int>(win[a][t][4 * G]), static_cast<int>(win[a][t][4 * G + 1]), static_cast<int>(win[a][t][4 * G + 2]),
                                        static_cast<int>(win[a][t][4 * G + 3]) };
                        d = __builtin_amdgcn_mfma_i32_16x16x64_i8(band[(slot - 4 * G + W) % W], b, d, 0, 0, 0);
                    }
                    const uint32_t pk = bx_pack(d, mul);
                    if (a + 1 < P) win[a + 1][t][slot] = pk; else res[t] = pk;
                }
            }
            if (s + VPF < S) issue(s + VPF, cur);
            if (s >= FILL) {
                const int row = ys + 16 * (s - FILL) + 4 * q + p;
                const bool rok = row < ye;
                const uint32_t off = static_cast<uint32_t>(row) * static_cast<uint32_t>(pitch) + col0;
#pragma unroll
                for (int t = 0; t < NT; ++t) res[t] = bx_quad_transpose(res[t], sel1, sel2);
                if (whole16) {
#pragma unroll
                    for (int k = 0; k < NT / 4; ++k) {
                        const u4 w = { res[4 * k], res[4 * k + 1], res[4 * k + 2], res[4 * k + 3] };
                        const bool ok = rok && col0 + 16 * k < static_cast<uint32_t>(pitch);
                        __builtin_amdgcn_raw_buffer_store_b128(w, rout, ok ? off + 16 * k : kDropped, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const bool ok = rok && col0 + 4 * t < static_cast<uint32_t>(pitch);
                        __builtin_amdgcn_raw_buffer_store_b32(res[t], rout, ok ? off + 4 * t : kDropped, 0, 0);
                    }
                }
            }
        }
    }
}

// developer knobs (tools/bx_dev.py sweeps them): BLUR_BX_VSEG / BLUR_BX_HSEG = segments per column strip / per row, BLUR_BX_VNT = 4 | 8
int bx_env(const char* name)
{
    const char* e = getenv(name);
    return e && *e ? atoi(e) : 0;
}

// band entry and multiplier for a box of n = 2 r + 1 taps (see the header); false: outside the exact range
bool bx_constants(int r, int* s_band, uint32_t* mul)
{
    const long long n = 2ll * r + 1;
    if (r < 1 || n >= 400) return false;
    const int s = n >= 5 ? 64 : 127;
    const unsigned long long m = ((1ull << 32) + static_cast<unsigned long long>(s * n) - 1) / static_cast<unsigned long long>(s * n);
    if (m >= (1ull << 24)) return false;
    *s_band = s;
    *mul = static_cast<uint32_t>(m);
    return true;
}

template <int NB, int P, int NT>
hipError_t bx_launch_vert(hipStream_t st, const uint8_t* in, uint8_t* out, int h, int pitch, int r, int s_band, uint32_t mul, int num_cus)
{
    constexpr int W = 4 * NB;
    const int nstrips = (pitch + 16 * NT - 1) / (16 * NT);
    // segments: four waves per SIMD in all (measured at 8K, k = 41, P = 3: 82 / 62 / 53 / 54 / 64 us with 3 / 6 / 10 / 14 / 30 segments
    // of 360 strips -- the walk is a chain of dependent matrix and vector instructions and lives on other waves' work), but no
    // segment shorter than twice the rows it takes to fill its pipeline
    const int fill_rows = 16 * P * (W - 1);
    int nseg = (16 * num_cus + nstrips / 2) / nstrips;
    const int most = h / (2 * fill_rows);
    if (nseg > most) nseg = most;
    if (bx_env("BLUR_BX_VSEG") > 0) nseg = bx_env("BLUR_BX_VSEG");
    if (nseg < 1) nseg = 1;
    int seg_rows = ((h + nseg - 1) / nseg + 15) / 16 * 16;
    nseg = (h + seg_rows - 1) / seg_rows;
    const int nwaves = nstrips * nseg;
    hipLaunchKernelGGL((bx_vert_kernel<NB, P, NT>), dim3((nwaves + 3) / 4), dim3(256), 0, st, in, out, h, pitch, r, s_band, mul, seg_rows, nstrips, nwaves);
    return hipGetLastError();
}

template <int NB, int NT>
hipError_t bx_launch_vert_p(hipStream_t st, const uint8_t* in, uint8_t* out, int h, int pitch, int r, int passes, int s_band, uint32_t mul, int num_cus)
{
    switch (passes) {
    case 1: return bx_launch_vert<NB, 1, NT>(st, in, out, h, pitch, r, s_band, mul, num_cus);
    case 2: return bx_launch_vert<NB, 2, NT>(st, in, out, h, pitch, r, s_band, mul, num_cus);
    default: return bx_launch_vert<NB, 3, NT>(st, in, out, h, pitch, r, s_band, mul, num_cus);
    }
}

// ---- horizontal ----------------------------------------------------------------------------------------------------------
// The same pipeline along the rows of interleaved pixels: positions are BYTES of a row, the band has its taps on every C-th byte
// (reach C r), a wave owns NT groups of 16 rows (lane n = row) and a segment of the row.  Lane group q loads 16 bytes (dwordx4) at
// byte 64 b + 16 q of the row for the four steps of block b; a 4 x 4 transpose of dwords across the four lane groups (two
// v_permlane32_swap + two v_permlane16_swap) turns them into "dword q of the 16 bytes of step 4 b + j" in register j, the window
// dword of that step.  Stores take the same road back, four steps at a time.  Bytes left of the row and right of it come from
// `margins`: per row the mirrored pixels left of the row followed by its first 16 bytes, then its last 16 bytes followed by the
// mirrored pixels right of it (bx_margins_kernel), so that a 16-byte group that leaves the row on either side is one load too.
struct BxHorzGeom {
    int h, pitch, C, r;
    int seg_bytes, nseg, ngroups;     // segment length (multiple of 64), segments per row, groups of NT * 16 rows
    int ml, mr, mpitch;               // margin bytes left / right of the row, bytes of margins per row (ml + 16 + 16 + mr)
};

constexpr int kMarginRows = 8;

__global__ __launch_bounds__(256) void bx_margins_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ margins, BxHorzGeom g)
{
    // one thread per margin byte of kMarginRows rows (blockIdx.y); positions are taken relative to the row's end they belong to, so
    // every division is of a small number by C = 1, 3 or 4 (multiply and shift)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= g.mpitch) return;
    const int w = g.pitch / g.C;
    const bool left = k < g.ml + 16;
    const int rel = (left ? k - g.ml : k - g.ml - 32) + g.C * 4096;            // byte offset from pixel 0 / from pixel w, made positive
    const int xq = g.C == 3 ? static_cast<int>((static_cast<uint32_t>(rel) * 43691u) >> 17) : g.C == 4 ? rel >> 2 : rel;      // rel < 2^16
    const int c = rel - xq * g.C;
    int x = xq - 4096 + (left ? 0 : w);
    for (int it = 0; it < 64 && (x < 0 || x >= w); ++it) {                      // reflect-101, as often as it takes
        x = x < 0 ? -x : x;
        x = x >= w ? 2 * (w - 1) - x : x;
    }
    x = w > 1 ? x : 0;
    const int r0 = blockIdx.y * kMarginRows, r1 = min(g.h, r0 + kMarginRows);
    uint8_t v[kMarginRows];
#pragma unroll
    for (int i = 0; i < kMarginRows; ++i) v[i] = in[static_cast<size_t>(min(r0 + i, g.h - 1)) * g.pitch + x * g.C + c];
#pragma unroll
    for (int i = 0; i < kMarginRows; ++i)
        if (r0 + i < r1) margins[static_cast<size_t>(r0 + i) * g.mpitch + k] = v[i];
}

__device__ __forceinline__ void bx_transpose_groups(uint32_t (&x)[4])
{
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 t;
    t = __builtin_amdgcn_permlane32_swap(x[0], x[2], false, false); x[0] = t[0]; x[2] = t[1];
    t = __builtin_amdgcn_permlane32_swap(x[1], x[3], false, false); x[1] = t[0]; x[3] = t[1];
    t = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false); x[0] = t[0]; x[1] = t[1];
    t = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false); x[2] = t[0]; x[3] = t[1];
}

template <int NB, int P, int NT>
__global__ __launch_bounds__(256) void bx_horz_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint8_t* __restrict__ margins, BxHorzGeom g,
                                                      int s_band, uint32_t mul, int nwaves)
{
    constexpr int W = 4 * NB, DELTA = 32 * NB - 8, FILL = (P * (W - 1) + 3) / 4 * 4;       // whole blocks of four steps
    const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * 4 + (threadIdx.x >> 6)));      // wave-uniform: scalar branches below
    if (wid >= nwaves) return;
    const int l = threadIdx.x & 63, n = l & 15, q = l >> 4;
    const int grp = wid % g.ngroups, seg = wid / g.ngroups;
    const int xs = seg * g.seg_bytes, xe = min(g.pitch, xs + g.seg_bytes);
    const int nblk = FILL / 4 + (xe - xs + 63) / 64;
    // sweep a's tile of step s covers bytes X0 - a DELTA + 16 s ..; the last sweep's tile of step FILL starts at xs
    const int X0 = xs + P * DELTA - 16 * FILL;
    const uint32_t obytes = static_cast<uint32_t>(g.h) * static_cast<uint32_t>(g.pitch);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, obytes, kRsrcWord3);

    v4i band[W];
#pragma unroll
    for (int u = 0; u < W; ++u) band[u] = bx_band<NB>(u, n, q, g.r, g.C, s_band);
    const int nt = 2 * g.r + 1, bias_last = s_band * (128 * nt + g.r), bias_mid = s_band * (256 * nt + g.r);
    const v4i cin_last = { bias_last, bias_last, bias_last, bias_last }, cin_mid = { bias_mid, bias_mid, bias_mid, bias_mid };

    const uint8_t* irow[NT];
    const uint8_t* mrow[NT];
    uint32_t orow[NT];
    bool rok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int row = (grp * NT + t) * 16 + n;
        rok[t] = row < g.h;
        const int rc = rok[t] ? row : g.h - 1;
        irow[t] = in + static_cast<size_t>(rc) * g.pitch;
        mrow[t] = margins + static_cast<size_t>(rc) * g.mpitch;
        orow[t] = static_cast<uint32_t>(rc) * static_cast<uint32_t>(g.pitch);
    }

    uint32_t win[P][NT][W];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int d = 0; d < W; ++d) win[a][t][d] = 0;

    u4 ld[2][NT];
    auto issue = [&](int b, u4 (&dst)[NT]) __attribute__((always_inline)) {
        const int pos = X0 + 64 * b + 16 * q;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint8_t* src = pos < 0 ? mrow[t] + (pos + g.ml) : pos + 16 > g.pitch ? mrow[t] + (g.ml + 16 + pos - (g.pitch - 16)) : irow[t] + pos;
            dst[t] = *reinterpret_cast<const u4*>(src);
        }
    };
    issue(0, ld[0]);
    if (nblk > 1) issue(1, ld[1]);

    for (int b0 = 0; b0 < nblk; b0 += 2 * NB) {
#pragma unroll
        for (int bb = 0; bb < 2 * NB; ++bb) {
            const int b = b0 + bb;
            if (b >= nblk) break;
            u4 (&cur)[NT] = ld[bb & 1];
            uint32_t x[NT][4], y[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[t][j] = cur[t][j];
                bx_transpose_groups(x[t]);
            }
            if (b + 2 < nblk) issue(b + 2, cur);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = (4 * bb + j) % W;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    win[0][t][slot] = x[t][j] ^ 0x80808080u;
#pragma unroll
                    for (int a = 0; a < P; ++a) {
                        v4i d = a + 1 < P ? cin_mid : cin_last;
#pragma unroll
                        for (int G = 0; G < NB; ++G) {
                            const v4i bop = { static_cast<int>(win[a][t][4 * G]), static_cast<int>(win[a][t][4 * G + 1]), static_cast<int>(win[a][t][4 * G + 2]),
                                              static_cast<int>(win[a][t][4 * G + 3]) };
                            d = __builtin_amdgcn_mfma_i32_16x16x64_i8(band[(slot - 4 * G + W) % W], bop, d, 0, 0, 0);
                        }
                        const uint32_t pk = bx_pack(d, mul);
                        if (a + 1 < P) win[a + 1][t][slot] = pk; else y[t][j] = pk;
                    }
                }
            }
            if (4 * b >= FILL) {
                const int pos = xs + 64 * (b - FILL / 4) + 16 * q;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bx_transpose_groups(y[t]);
                    if (pos + 16 <= xe) {
                        const u4 wv = { y[t][0], y[t][1], y[t][2], y[t][3] };
                        __builtin_amdgcn_raw_buffer_store_b128(wv, rout, rok[t] ? orow[t] + pos : kDropped, 0, 0);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            __builtin_amdgcn_raw_buffer_store_b32(y[t][k], rout, (rok[t] && pos + 4 * k + 4 <= xe) ? orow[t] + pos + 4 * k : kDropped, 0, 0);
                    }
                }
            }
        }
    }
}

template <int NB> constexpr int bx_horz_fill(int P) { return (P * (4 * NB - 1) + 3) / 4 * 4; }

template <int NB, int P, int NT>
hipError_t bx_launch_horz(hipStream_t st, const uint8_t* in, uint8_t* out, uint8_t* margins, BxHorzGeom g, int s_band, uint32_t mul, int num_cus)
{
    constexpr int FILL = bx_horz_fill<NB>(P);
    g.ngroups = (g.h + 16 * NT - 1) / (16 * NT);
    // segments: four waves per SIMD in all (8K, k = 41, P = 3: 100 / 75 / 70 / 69 / 80 / 86 us with 4 / 6 / 10 / 15 / 20 / 30 segments of
    // 270 row groups; two groups of 16 rows per wave and half the waves: 84 .. 108 us), none shorter than twice the bytes it takes to
    // fill its pipeline
    int nseg = (16 * num_cus + g.ngroups / 2) / g.ngroups;
    const int most = g.pitch / (2 * 16 * FILL);
    if (nseg > most) nseg = most;
    if (bx_env("BLUR_BX_HSEG") > 0) nseg = bx_env("BLUR_BX_HSEG");
    if (nseg < 1) nseg = 1;
    g.seg_bytes = ((g.pitch + nseg - 1) / nseg + 63) / 64 * 64;
    g.nseg = (g.pitch + g.seg_bytes - 1) / g.seg_bytes;
    hipLaunchKernelGGL(bx_margins_kernel, dim3((g.mpitch + 255) / 256, (g.h + kMarginRows - 1) / kMarginRows), dim3(256), 0, st, in, margins, g);
    const int nwaves = g.ngroups * g.nseg;
    hipLaunchKernelGGL((bx_horz_kernel<NB, P, NT>), dim3((nwaves + 3) / 4), dim3(256), 0, st, in, out, margins, g, s_band, mul, nwaves);
    return hipGetLastError();
}

template <int NB, int NT>
hipError_t bx_launch_horz_p(hipStream_t st, const uint8_t* in, uint8_t* out, uint8_t* margins, BxHorzGeom g, int passes, int s_band, uint32_t mul, int num_cus)
{
    switch (passes) {
    case 1: return bx_launch_horz<NB, 1, NT>(st, in, out, margins, g, s_band, mul, num_cus);
    case 2: return bx_launch_horz<NB, 2, NT>(st, in, out, margins, g, s_band, mul, num_cus);
    default: return bx_launch_horz<NB, 3, NT>(st, in, out, margins, g, s_band, mul, num_cus);
    }
}

// ---- horizontal, three channels, by channel plane ------------------------------------------------------------------------
// Over interleaved RGB the band above has its taps on every third byte: two thirds of every matrix instruction multiply zeros,
// and a box of 41 taps needs a window of three instructions (reach 60 bytes) where 20 PIXELS of reach fit one.  Here a step is 16
// pixels (48 bytes of a row): lane (n = row, q) loads the 12 bytes of pixels 4 q .. 4 q + 3 itself (a wave's load = 16 rows x 48
// contiguous bytes; no transposes across the lane groups), six v_perm_b32 split them into the three channels' window dwords --
// exactly the layout of the vertical kernel with the channels in the place of its column tiles -- and six more put the three
// result dwords back together as 12 bytes.  The few steps that reach over an end of the row gather their mirrored pixels one by one.  Per 16 pixels and sweep: 3 NB instructions instead of 3 NB' with NB' = ceil((3 r +
// 8) / 32) >= 2 NB.  Any width and any alignment (12-byte loads and stores at byte addresses work on gfx950:
// tools/probes/unaligned_probe.hip); a quad of pixels cut by the end of the row leaves byte by byte.
template <int NB, int P>
__global__ __launch_bounds__(256) void bx_horz3_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, BxHorzGeom g, int s_band, uint32_t mul, int nwaves)
{
#ifndef BX_PF
#define BX_PF 4
#endif
    constexpr int W = 4 * NB, DELTA = 32 * NB - 8, FILL = P * (W - 1), PF = BX_PF;      // PF: steps of loads in flight
    static_assert((2 * W) % PF == 0, "the prefetch ring is indexed statically");
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * 4 + (threadIdx.x >> 6)));      // wave-uniform: scalar branches below
    if (wid >= nwaves) return;
    const int l = threadIdx.x & 63, n = l & 15, q = l >> 4;
    const int grp = wid % g.ngroups, seg = wid / g.ngroups;
    const int w = g.pitch / 3, seg_px = g.seg_bytes / 3;
    const int xs = seg * seg_px, xe = min(w, xs + seg_px);
    const int S = FILL + (xe - xs + 15) / 16;
    // sweep a's tile of step s covers pixels X0 - a DELTA + 16 s ..; the last sweep's tile of step FILL starts at xs
    const int X0 = xs + P * DELTA - 16 * FILL;
    const uint32_t obytes = static_cast<uint32_t>(g.h) * static_cast<uint32_t>(g.pitch);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, obytes, kRsrcWord3);

    v4i band[W];
#pragma unroll
    for (int u = 0; u < W; ++u) band[u] = bx_band<NB>(u, n, q, g.r, 1, s_band);
    const int nt = 2 * g.r + 1, bias_last = s_band * (128 * nt + g.r), bias_mid = s_band * (256 * nt + g.r);
    const v4i cin_last = { bias_last, bias_last, bias_last, bias_last }, cin_mid = { bias_mid, bias_mid, bias_mid, bias_mid };

    const int row = grp * 16 + n;
    const bool rok = row < g.h;
    const int rc = rok ? row : g.h - 1;
    const uint8_t* irow = in + static_cast<size_t>(rc) * g.pitch;
    const uint32_t orow = static_cast<uint32_t>(rc) * static_cast<uint32_t>(g.pitch);

    uint32_t win[P][3][W];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < W; ++d) win[a][c][d] = 0;

    // a step that lies inside the row (wave-uniform test) is one buffer load / store with the step in the scalar offset: no vector
    // arithmetic per step; steps that touch a margin or the cut end of the row pick their addresses per lane
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in), 0, obytes, kRsrcWord3);
    const uint32_t vin = orow + 12u * q, vout = rok ? orow + 12u * q : kDropped;
    u3 ld[PF];
    auto issue = [&](int s, u3& dst) __attribute__((always_inline)) {
        const int p0 = 3 * (X0 + 16 * s);                            // byte of the row where the step starts (wave-uniform)
        if (p0 >= 0 && p0 + 48 <= g.pitch) {
            dst = __builtin_amdgcn_raw_buffer_load_b96(rin, vin, p0, 0);
        } else {
            // a step that reaches over an end of the row (a few per row, in the first and last segment).  A quad of pixels that lies
            // wholly beyond an end is the pixel-reversed copy of four adjacent pixels of the row: one 12-byte load and the reversal;
            // quads cut by the end (widths that are no multiple of 4) or mirrored more than once (tiny rows) go pixel by pixel
            const int X = X0 + 16 * s + 4 * q;
            const bool lo = X + 3 < 0, hi = X >= w, mir = lo || hi;
            const int xq = lo ? -X - 3 : (hi ? 2 * (w - 1) - X - 3 : X);
            const bool simple = xq >= 0 && xq + 3 < w;
            if (!__any(!simple)) {
                const u3 t = *reinterpret_cast<const u3*>(irow + 3 * xq);
                // pixels 3, 2, 1, 0 of the loaded four: bytes [9 10 11 6] [7 8 3 4] [5 0 1 2]
                const uint32_t r0 = __builtin_amdgcn_perm(t[2], t[1], 0x02070605u);
                const uint32_t r1 = (t[1] >> 24) | ((t[2] & 0xffu) << 8) | ((t[0] >> 24) << 16) | ((t[1] & 0xffu) << 24);
                const uint32_t r2 = __builtin_amdgcn_perm(t[1], t[0], 0x02010005u);
                dst[0] = mir ? r0 : t[0]; dst[1] = mir ? r1 : t[1]; dst[2] = mir ? r2 : t[2];
            } else {
                uint32_t b[12];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int x = X + k;
                    for (int it = 0; it < 64 && (x < 0 || x >= w); ++it) {
                        x = x < 0 ? -x : x;
                        x = x >= w ? 2 * (w - 1) - x : x;
                    }
                    const uint8_t* px = irow + 3 * x;
                    b[3 * k] = px[0]; b[3 * k + 1] = px[1]; b[3 * k + 2] = px[2];
                }
                dst[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                dst[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                dst[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
            }
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (k < S) issue(k, ld[k]);

    for (int s0 = 0; s0 < S; s0 += 2 * W) {
#pragma unroll
        for (int uu = 0; uu < 2 * W; ++uu) {
            const int s = s0 + uu, slot = uu % W;
            if (s >= S) break;
            const u3 cur = ld[uu % PF];
            if (s + PF < S) issue(s + PF, ld[uu % PF]);
            // pixels p = 0 .. 3, channel c = byte 3 p + c of the 12: -> one dword per channel (signed: x ^ 0x80)
            const uint32_t d0 = cur[0] ^ 0x80808080u, d1 = cur[1] ^ 0x80808080u, d2 = cur[2] ^ 0x80808080u;
            uint32_t ch[3];
            ch[0] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x00060300u), 0x05020100u);      // bytes 0, 3, 6, 9
            ch[1] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x00070401u), 0x06020100u);      // bytes 1, 4, 7, 10
            ch[2] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x00000502u), 0x07040100u);      // bytes 2, 5, 8, 11
            uint32_t res[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) win[0][c][slot] = ch[c];
#pragma unroll
            for (int a = 0; a < P; ++a) {
                // the three channels' products are independent of each other: issued together, packed together (one pair of waits)
                v4i d[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    d[c] = a + 1 < P ? cin_mid : cin_last;
#pragma unroll
                    for (int G = 0; G < NB; ++G) {
                        const v4i bop = { static_cast<int>(win[a][c][4 * G]), static_cast<int>(win[a][c][4 * G + 1]), static_cast<int>(win[a][c][4 * G + 2]),
                                          static_cast<int>(win[a][c][4 * G + 3]) };
                        d[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(band[(slot - 4 * G + W) % W], bop, d[c], 0, 0, 0);
                    }
                }
                // (the SDWA form of the multiply -- 4 instructions per dword instead of 7, its waits carried inside the inline assembly --
                // gave equal bytes and equal time here, 52.2 against 52.7 us at 8K: the kernel waits on memory, so the plain form stays)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t pk = bx_pack(d[c], mul);
                    if (a + 1 < P) win[a + 1][c][slot] = pk; else res[c] = pk;
                }
            }
            if (s >= FILL) {
                const int px = xs + 16 * (s - FILL) + 4 * q;
                // res[c] = channel c of pixels px .. px + 3 -> 12 interleaved bytes
                const uint32_t X = __builtin_amdgcn_perm(res[1], res[0], 0x05010400u);      // [r0.0, r1.0, r0.1, r1.1]
                const uint32_t Y = __builtin_amdgcn_perm(res[1], res[0], 0x07030602u);      // [r0.2, r1.2, r0.3, r1.3]
                const uint32_t w0 = __builtin_amdgcn_perm(res[2], X, 0x02040100u);          // [X0, X1, r2.0, X2]
                const uint32_t Z = __builtin_amdgcn_perm(res[2], X, 0x0c0c0503u);           // [X3, r2.1, 0, 0]
                const uint32_t w1 = __builtin_amdgcn_perm(Y, Z, 0x05040100u);               // [Z0, Z1, Y0, Y1]
                const uint32_t w2 = __builtin_amdgcn_perm(res[2], Y, 0x07030206u);          // [r2.2, Y2, Y3, r2.3]
                const uint32_t off = orow + 3u * static_cast<uint32_t>(px);
                const u3 wv = { w0, w1, w2 };
                const int p0 = xs + 16 * (s - FILL);                    // (wave-uniform)
                if (p0 + 16 <= xe) {
                    __builtin_amdgcn_raw_buffer_store_b96(wv, rout, vout, 3 * p0, 0);
                } else {                                                // the row's last, cut step: a cut quad leaves as single bytes
                    __builtin_amdgcn_raw_buffer_store_b96(wv, rout, (rok && px + 4 <= xe) ? off : kDropped, 0, 0);
                    const int tail = (rok && px < xe && px + 4 > xe) ? 3 * (xe - px) : 0;
#pragma unroll
                    for (int b = 0; b < 9; ++b) {
                        const uint32_t word = b < 4 ? w0 : (b < 8 ? w1 : w2);
                        __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(word >> (8 * (b & 3))), rout, b < tail ? off + b : kDropped, 0, 0);
                    }
                }
            }
        }
    }
}

template <int NB, int P>
hipError_t bx_launch_horz3(hipStream_t st, const uint8_t* in, uint8_t* out, BxHorzGeom g, int s_band, uint32_t mul, int num_cus)
{
    constexpr int FILL = P * (4 * NB - 1);
    const int w = g.pitch / 3;
    g.ngroups = (g.h + 15) / 16;
    // segments of a row: waves for every SIMD, none shorter than twice the pixels it takes to fill its pipeline
    // (8K, k = 41, P = 3, 270 row groups: 63 / 49 / 55 / 49 / 55 / 51 / 56 us with 6 / 7 / 9 / 11 / 13 / 15 / 22 segments: the waves should come to
    // just under a whole number per SIMD -- 2 (7 segments) and 3 (11) do, 2.4 (9) and 3.4 (13) leave SIMDs waiting for the others)
    int nseg = (bx_env("BLUR_BX_HWAVES") > 0 ? bx_env("BLUR_BX_HWAVES") : 12) * num_cus / g.ngroups;
    const int most = w / (2 * 16 * FILL);
    if (nseg > most) nseg = most;
    if (bx_env("BLUR_BX_HSEG") > 0) nseg = bx_env("BLUR_BX_HSEG");
    if (nseg < 1) nseg = 1;
    const int seg_px = ((w + nseg - 1) / nseg + 15) / 16 * 16;
    g.seg_bytes = 3 * seg_px;
    g.nseg = (w + seg_px - 1) / seg_px;
    const int nwaves = g.ngroups * g.nseg;
    hipLaunchKernelGGL((bx_horz3_kernel<NB, P>), dim3((nwaves + 3) / 4), dim3(256), 0, st, in, out, g, s_band, mul, nwaves);
    return hipGetLastError();
}

template <int NB>
hipError_t bx_launch_horz3_p(hipStream_t st, const uint8_t* in, uint8_t* out, BxHorzGeom g, int passes, int s_band, uint32_t mul, int num_cus)
{
    switch (passes) {
    case 1: return bx_launch_horz3<NB, 1>(st, in, out, g, s_band, mul, num_cus);
    case 2: return bx_launch_horz3<NB, 2>(st, in, out, g, s_band, mul, num_cus);
    default: return bx_launch_horz3<NB, 3>(st, in, out, g, s_band, mul, num_cus);
    }
}

// geometry of the channel-plane kernel: three channels, r <= 56, rows of at least 128 bytes
bool bx_horz3_geom(int h, int w, int C, int r, int passes, BxHorzGeom* g)
{
    if (C != 3 || r < 1 || r > 56 || passes < 1 || passes > 3 || bx_env("BLUR_BX_NO_HORZ3")) return false;
    const long long pitch = 3ll * w;
    if (pitch < 128 || pitch * h >= (1ll << 31)) return false;
    g->h = h; g->pitch = static_cast<int>(pitch); g->C = 3; g->r = r;
    g->ml = g->mr = g->mpitch = 0;                      // (no margins: the kernel mirrors the row's ends itself)
    g->seg_bytes = g->nseg = g->ngroups = 0;
    return true;
}

// window blocks of 64 bytes for a reach of C r bytes; 0: wider than the instantiated kernels
int bx_horz_blocks(int C, int r)
{
    for (int nb = 1; nb <= 4; ++nb)
        if (C * r <= 32 * nb - 8) return nb;
    return 0;
}

bool bx_horz_geom(int h, int w, int C, int r, int passes, BxHorzGeom* g)
{
    const int nb = bx_horz_blocks(C, r);
    if (!nb || passes < 1 || passes > 3 || !(C == 1 || C == 3 || C == 4)) return false;
    const long long pitch = static_cast<long long>(w) * C;
    if ((pitch & 3) || pitch < 128 || pitch * h >= (1ll << 31)) return false;
    const int fill = (passes * (4 * nb - 1) + 3) / 4 * 4, delta = 32 * nb - 8;
    g->h = h; g->pitch = static_cast<int>(pitch); g->C = C; g->r = r;
    g->ml = 16 * fill - passes * delta;                 // -X0 of the first segment
    g->mr = passes * delta + 16 + 64 + 16;              // the last segment's last block ends at most this far past the row
    g->mpitch = g->ml + 16 + 16 + g->mr;
    g->seg_bytes = g->nseg = g->ngroups = 0;
    return true;
}

}  // namespace

hipError_t bx_vertical(hipStream_t st, const uint8_t* in, uint8_t* out, int h, int pitch, int r, int passes, int num_cus, bool* ran)
{
    *ran = false;
    int s_band = 0;
    uint32_t mul = 0;
    if (passes < 1 || passes > 3 || !bx_constants(r, &s_band, &mul)) return hipSuccess;
    const int nb = r <= 24 ? 1 : r <= 56 ? 2 : 0;
    if (!nb) return hipSuccess;
    const int delta = 32 * nb - 8;
    if ((pitch & 3) || (reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3)) return hipSuccess;
    if (h < passes * delta + 32 || static_cast<long long>(h) * pitch >= (1ll << 31)) return hipSuccess;
    *ran = true;
    // narrow images: 64-byte strips give twice the waves
    const bool narrow = bx_env("BLUR_BX_VNT") ? bx_env("BLUR_BX_VNT") == 4 : pitch < 128 * 2 * num_cus;
    if (nb == 1) return narrow ? bx_launch_vert_p<1, 4>(st, in, out, h, pitch, r, passes, s_band, mul, num_cus) : bx_launch_vert_p<1, 8>(st, in, out, h, pitch, r, passes, s_band, mul, num_cus);
    return narrow ? bx_launch_vert_p<2, 4>(st, in, out, h, pitch, r, passes, s_band, mul, num_cus) : bx_launch_vert_p<2, 8>(st, in, out, h, pitch, r, passes, s_band, mul, num_cus);
}

size_t bx_horizontal_scratch(int h, int w, int C, int r, int passes)
{
    BxHorzGeom g;
    if (bx_horz3_geom(h, w, C, r, passes, &g)) return 64;          // (the channel-plane kernel needs none)
    if (!bx_horz_geom(h, w, C, r, passes, &g)) return 0;
    return static_cast<size_t>(h) * g.mpitch + 64;
}

hipError_t bx_horizontal(hipStream_t st, const uint8_t* in, uint8_t* out, uint8_t* margins, int h, int w, int C, int r, int passes, int num_cus, bool* ran)
{
    *ran = false;
    int s_band = 0;
    uint32_t mul = 0;
    BxHorzGeom g;
    if (bx_constants(r, &s_band, &mul) && bx_horz3_geom(h, w, C, r, passes, &g)) {
        *ran = true;
        return r <= 24 ? bx_launch_horz3_p<1>(st, in, out, g, passes, s_band, mul, num_cus) : bx_launch_horz3_p<2>(st, in, out, g, passes, s_band, mul, num_cus);
    }
    if (!margins || !bx_constants(r, &s_band, &mul) || !bx_horz_geom(h, w, C, r, passes, &g)) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3) || (reinterpret_cast<uintptr_t>(margins) & 3)) return hipSuccess;
    *ran = true;
    switch (bx_horz_blocks(C, r)) {
    case 1: return bx_launch_horz_p<1, 1>(st, in, out, margins, g, passes, s_band, mul, num_cus);
    case 2: return bx_launch_horz_p<2, 1>(st, in, out, margins, g, passes, s_band, mul, num_cus);
    case 3: return bx_launch_horz_p<3, 1>(st, in, out, margins, g, passes, s_band, mul, num_cus);
    default: return bx_launch_horz_p<4, 1>(st, in, out, margins, g, passes, s_band, mul, num_cus);
    }
}

}  // namespace blur_amd
