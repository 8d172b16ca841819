// mx_kernels.hpp -- the separable blur as two banded Toeplitz products on the f16 matrix cores (gfx950).
//
// Why this exists beside the FFT kernels: both FFT families are bound by vector-ALU issue, not by HBM (DESIGN.md §8.1:
// 90 % / 68 % VALU busy at 0.26 of the HBM roofline, and the butterfly arithmetic alone is above the 70 % budget).  For
// the kernel widths the reference is used with (sigma 20: 131 taps) the same linear map
//
//     out[line][o] = sum_t taps[t] * in[line][o + t]                 (inside the crop, Source.cpp:536,558: the circular
//                                                                     FFT product never wraps into a kept pixel)
//
// costs 2 x 176 multiply-adds per output as a dense 32 x 176 Toeplitz tile -- 3 % of the chip's f16 MFMA rate -- so the
// path becomes what SURVEY 8(d) assumed it was: a stream over HBM.  Precision: the u8 image is exact in binary16; the
// taps are split into hi + lo halves (22 significant bits); the intermediate V (24-bit fixed point in memory) is split
// into hi + lo halves when the column pass decodes it; products are exact in the f32 accumulators of
// v_mfma_f32_32x32x16_f16.  The Nyquist-slot quirk of pffft_() (Source.cpp:420-425) is a rank-one term per line, made of
// exact integer alternating sums of the image that the row kernel leaves behind as partial sums (notes further down).
//
//   mx_rowpass_u8   u8 BGR image -> V (24-bit fixed point, [strip][8 rows][lane][24 bytes], see MxV24x8)
//       unit = 32 rows x 128 pixels: reflect-101 + deinterleave + u8 -> f16 into LDS, A = data (32 rows x 16 window
//       positions), B = Toeplitz fragment (registers), D = 32 rows x 32 outputs; re-interleaved through LDS, packed to
//       24 bits and stored as 24-byte operand fragments
//   mx_colpass_u8   V -> u8 BGR image
//       a wave owns 32 adjacent values of V's rows and walks down the image: each 16-row block is loaded ONCE (two loads:
//       the lane's 8 consecutive rows are 24 contiguous bytes), decoded, split into hi + lo halves, and multiplied into
//       the (NKB + 1) / 2 output tiles whose windows contain it (A = Toeplitz fragment of the block's offset in that
//       tile's window).  No LDS, no barrier; channels never need separating because the convolution runs along rows of
//       V and every interleaved column is independent.
//   mx_quirk_rows, mx_quirk_cols   the quirk's two term vectors from the row kernel's partial sums
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace blur_amd {

typedef _Float16 mx_half8 __attribute__((ext_vector_type(8)));
typedef float mx_float16 __attribute__((ext_vector_type(16)));

constexpr int kMxRowChunk = 128;        // pixels per row-pass unit
constexpr int kMxStagePitch = 392;      // floats per staged row (384 + 8: the two half-waves land in disjoint banks)
constexpr float kMxUnscale = 1.f / 16384.f;   // 2^-kMxScaleLog2 (host_math.hpp)

// The intermediate V between the two kernels is 24-bit fixed point: q = round(v * 2^16), three bytes per value.  Range
// [0, 256): the taps are non-negative with sum <= 1 (the engine refuses other kernels), so the row pass gives 0..255.13;
// resolution 1.5e-5 grey levels, i.e. a uniform error of at most 7.6e-6 that the column pass averages over ~70 rows.
// (The row pass's quirk term, up to +-255, is added by the column pass when it decodes V.)  A quarter less traffic on the
// 12 of every 15 B/px that are V.  Layout: [frame][strip of 32 values of a V row][group of 8 rows][lane 0..31][24 bytes]
// -- the 24 bytes are the lane's 8 consecutive rows, exactly its B-operand fragment of the column pass: two loads per
// 16-row block instead of eight, 768 contiguous bytes per (strip, row group).
constexpr float kMxV24Scale = 65536.f, kMxV24Offset = 0.f;
struct __attribute__((packed, aligned(8))) MxV24x8 { uint32_t d[6]; };

__device__ __forceinline__ MxV24x8 mx_v24_pack(const float (&v)[8])
{
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)      // (the clamp only ever acts on columns past the image, whose values are never used)
        q[k] = static_cast<uint32_t>(__builtin_rintf(fminf(fmaxf(__builtin_fmaf(v[k], kMxV24Scale, kMxV24Offset * kMxV24Scale), 0.f), 16777215.f)));
    MxV24x8 r;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        r.d[3 * t] = q[4 * t] | (q[4 * t + 1] << 24);
        r.d[3 * t + 1] = (q[4 * t + 1] >> 8) | (q[4 * t + 2] << 16);
        r.d[3 * t + 2] = (q[4 * t + 2] >> 16) | (q[4 * t + 3] << 8);
    }
    return r;
}

__device__ __forceinline__ void mx_v24_unpack(const uint32_t (&d)[6], float (&v)[8])
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t q0 = d[3 * t] & 0xffffffu;
        const uint32_t q1 = __builtin_amdgcn_alignbit(d[3 * t + 1], d[3 * t], 24) & 0xffffffu;
        const uint32_t q2 = __builtin_amdgcn_alignbit(d[3 * t + 2], d[3 * t + 1], 16) & 0xffffffu;
        const uint32_t q3 = d[3 * t + 2] >> 8;
        v[4 * t] = __builtin_fmaf(static_cast<float>(q0), 1.f / kMxV24Scale, -kMxV24Offset);
        v[4 * t + 1] = __builtin_fmaf(static_cast<float>(q1), 1.f / kMxV24Scale, -kMxV24Offset);
        v[4 * t + 2] = __builtin_fmaf(static_cast<float>(q2), 1.f / kMxV24Scale, -kMxV24Offset);
        v[4 * t + 3] = __builtin_fmaf(static_cast<float>(q3), 1.f / kMxV24Scale, -kMxV24Offset);
    }
}

struct MxGeom {
    int rows, cols, pad;
    int vpitch;       // floats per row of V: 3 cols rounded up to 32
    int nframes;
    int aligned;      // 1: 12-byte pixel groups of the source can be read as three aligned dwords
    int vrows;        // rows of V per frame: the image rows with their reflect-101 border ABOVE AND BELOW already in place
                      // (row re of V = image row refl(re - PADA)), so the column pass never reflects: 32 ntiles + 2 PADA
    int nxcd;         // XCDs of the device (hipDeviceAttributeNumberOfXccs): workgroup b runs on XCD b % nxcd
};
__host__ __device__ constexpr int mx_vrows(int rows, int nkb) { return 32 * ((rows + 31) / 32) + 32 * ((16 * (nkb - 2) + 31) / 32); }

// LDS pitch (halfs) of one row of the row-pass window: the dword pitch is 4 mod 8, so 16 lanes reading 16 bytes from 16
// consecutive rows touch 64 distinct banks
__host__ __device__ constexpr int mx_row_pitch(int nkb)
{
    int dw = (kMxRowChunk + 16 * (nkb - 2)) / 2;
    while ((dw & 7) != 4) ++dw;
    return 2 * dw;
}
__host__ __device__ constexpr size_t mx_row_lds(int nkb)
{
    const size_t in = static_cast<size_t>(3) * 32 * mx_row_pitch(nkb) * 2, stage = static_cast<size_t>(32) * kMxStagePitch * 4;
    return in > stage ? in : stage;
}

__device__ __forceinline__ int mx_refl(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);      // beyond one reflection only zero taps read it: any valid address will do
}

__device__ __forceinline__ uint32_t mx_pk(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));   // exact: integers 0..255
}

// ---------------------------------------------------------------------------------------------------------------
// raw pixels of one row-pass unit.  Thread t owns row t >> 3 of the unit's 32 rows and the groups of 4 pixels
// (t & 7) + 8 k of that row's window, k = 0..PER-1: every address is one per-thread base plus a constant, in global
// memory (12 bytes per group) and in LDS (8 bytes per group and channel)
template <int NKB> struct MxRowRaw {
    static constexpr int PADA = 8 * (NKB - 2), WIN = kMxRowChunk + 2 * PADA, GPR = WIN / 4, PER = (GPR + 7) / 8;
    uint32_t d[PER][3];
};

template <int NKB>
__device__ __forceinline__ void mx_row_issue(MxRowRaw<NKB>& raw, const uint8_t* __restrict__ src, const MxGeom& g, int u, int chunks, int rblocks, int tid)
{
    using R = MxRowRaw<NKB>;
    const int xc = u % chunks, rb = (u / chunks) % rblocks, f = u / (chunks * rblocks);
    const int x0 = xc * kMxRowChunk, r0 = rb * 32 - R::PADA;          // first image row of the unit's 32 rows of V (mirrored outside)
    const uint8_t* img = src + static_cast<size_t>(f) * g.rows * g.cols * 3;
    const int row = tid >> 3, g0 = tid & 7;
    const int r = mx_refl(r0 + row, g.rows);
    // interior unit (uniform): every group is whole, inside the image and dword aligned -> branch-free loads the compiler
    // can issue back to back; otherwise per-pixel reflect-101 (the two edge chunks of a row, odd widths)
    const bool interior = g.aligned && x0 - R::PADA >= 0 && x0 + kMxRowChunk + R::PADA <= g.cols;
    if (interior) {
        // uniform base + 32-bit lane offset (a frame is < 4 GiB)
        const uint32_t off = (static_cast<uint32_t>(r) * g.cols + static_cast<uint32_t>(x0 - R::PADA + 4 * g0)) * 3u;
        const uint8_t* p = img + off;
#pragma unroll
        for (int k = 0; k < R::PER; ++k) {
            typedef uint32_t u3 __attribute__((ext_vector_type(3)));
            const bool in = (R::GPR % 8 == 0) || k < R::PER - 1 || g0 < R::GPR % 8;
            const u3 t = *reinterpret_cast<const u3*>(in ? p + 96 * k : p);
            raw.d[k][0] = t[0]; raw.d[k][1] = t[1]; raw.d[k][2] = t[2];
        }
    } else {
        const uint8_t* line = img + static_cast<size_t>(r) * g.cols * 3;
#pragma unroll                                                        // (a run-time k would put raw.d[] into scratch memory)
        for (int k = 0; k < R::PER; ++k) {
            const int xg = x0 - R::PADA + 4 * (g0 + 8 * k);
            uint32_t b[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint8_t* px = line + 3 * mx_refl(xg + q, g.cols);
                b[3 * q] = px[0]; b[3 * q + 1] = px[1]; b[3 * q + 2] = px[2];
            }
            raw.d[k][0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            raw.d[k][1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            raw.d[k][2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        }
    }
}

// weight of element i of a line of `len` elements in the alternating sum of its reflect-101 padded line (see the quirk notes below)
__device__ __forceinline__ int mx_alt_weight(int i, int len, int pad)
{
    const int w = 1 + ((i >= 1 && i <= pad) ? 1 : 0) + ((i >= len - 1 - pad && i <= len - 2) ? 1 : 0);
    return ((i + pad) & 1) ? -w : w;
}

// sum over the 8 lanes that share a row (lane & 7 = group index): quad xor 1, quad xor 2, then the mirror of the 8-lane half
__device__ __forceinline__ float mx_sum8(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    return v;
}

// deinterleave + u8 -> binary16 + write to the LDS window [c][row][PW].  QUIRK: also the alternating sum of the unit's OWN
// 128 pixels of the thread's row per channel (weights of mx_alt_weight; exact in float: integers below 2^24), summed over
// the row's 8 threads: s[c], valid in every lane.  `plain`: no pixel of the unit is mirrored (weights are +-1 by parity).
template <int NKB, bool QUIRK>
__device__ __forceinline__ void mx_row_commit(const MxRowRaw<NKB>& raw, _Float16* in, int tid, const MxGeom& g, int x0, bool plain, float (&s)[3])
{
    using R = MxRowRaw<NKB>;
    constexpr int PW = mx_row_pitch(NKB), GI0 = R::PADA / 4;          // first group of the unit's own pixels
    const int row = tid >> 3, g0 = tid & 7;
    _Float16* base = in + row * PW + 4 * g0;
    s[0] = s[1] = s[2] = 0.f;
    const float flip = (g.pad & 1) ? -1.f : 1.f;
#pragma unroll
    for (int k = 0; k < R::PER; ++k) {
        if ((R::GPR % 8 == 0) || k < R::PER - 1 || g0 < R::GPR % 8) {
            const int grp = g0 + 8 * k;
            const bool own = QUIRK && grp >= GI0 && grp < GI0 + 32;
            float w[4] = { flip, -flip, flip, -flip };                  // x = x0 + 4 (grp - GI0) + q: (x + pad) & 1 = (q + pad) & 1
            if (QUIRK && !plain) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int x = x0 + 4 * (grp - GI0) + q;
                    w[q] = x < g.cols ? static_cast<float>(mx_alt_weight(x, g.cols, g.pad)) : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int byte = 3 * q + c;
                    v[q] = static_cast<float>((raw.d[k][byte >> 2] >> (8 * (byte & 3))) & 0xffu);
                }
                if (own) s[c] += (v[0] * w[0] + v[1] * w[1]) + (v[2] * w[2] + v[3] * w[3]);
                uint2 wd;
                wd.x = mx_pk(v[0], v[1]);
                wd.y = mx_pk(v[2], v[3]);
                *reinterpret_cast<uint2*>(base + c * 32 * PW + 32 * k) = wd;
            }
        }
    }
    if (QUIRK) { s[0] = mx_sum8(s[0]); s[1] = mx_sum8(s[1]); s[2] = mx_sum8(s[2]); }
}

// two workgroups per CU while the fragments (16 NKB registers) leave room for them.
// QUIRK: the kernel also leaves the partial sums the Nyquist-slot terms are made of (no extra pass over the image):
//   spart[f][chunk][image row][3]   alternating sum of the chunk's 128 pixels of that row (integers, exact)
//   vpart[f][row block of V][e]     sum over the block's image rows of wy(r) V[r][e] (float; e = 3 x + c, pitch vpitch)
template <int NKB, bool QUIRK>
__global__ __launch_bounds__(256, (NKB <= 17 ? 2 : 1)) void mx_rowpass_u8(const uint8_t* __restrict__ src, float* __restrict__ V, const mx_half8* __restrict__ frags, MxGeom g,
                                                        int chunks, int rblocks, int nunits, int* __restrict__ spart, float* __restrict__ vpart)
{
    constexpr int PW = mx_row_pitch(NKB), PADA = 8 * (NKB - 2);
    extern __shared__ __attribute__((aligned(16))) unsigned char mx_lds[];
    _Float16* in = reinterpret_cast<_Float16*>(mx_lds);      // [3][32][PW]
    float* stage = reinterpret_cast<float*>(mx_lds);         // [32][kMxStagePitch], after the MFMAs have read `in`
    float* wyv = reinterpret_cast<float*>(mx_lds + mx_row_lds(NKB));   // [32]: wy of the unit's rows, 0 for mirrored rows (QUIRK)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;

    mx_half8 th[NKB], tl[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        th[kb] = frags[kb * 64 + lane];
        tl[kb] = frags[(NKB + kb) * 64 + lane];
    }
    // Workgroups b and b + nxcd (8 on MI355X) run on the same XCD and share its L2.  Give every XCD a contiguous band of units (units are
    // numbered chunk-fastest), so that the chunks left and right of a unit -- which read 2 PADA of its 128 + 2 PADA window
    // columns -- are the same XCD's work at about the same time: without this the window overlap was fetched from memory
    // 2.5 times (PMC: 489 MB read for 199 MB of frames, profiles/r02mx_pmc_counters.json).
    const int nx = g.nxcd, xcd = blockIdx.x % nx, per_xcd = (nunits + nx - 1) / nx, lanes_x = (gridDim.x + nx - 1 - xcd) / nx;     // workgroups of this XCD
    const int ubeg = xcd * per_xcd, uend = min(nunits, ubeg + per_xcd);
    MxRowRaw<NKB> raw;
    int u = ubeg + blockIdx.x / nx;
    const int ustep = lanes_x;
    if (u < uend) mx_row_issue<NKB>(raw, src, g, u, chunks, rblocks, tid);
    for (; u < uend; u += ustep) {
        const int xc = u % chunks, rb = (u / chunks) % rblocks, f = u / (chunks * rblocks);
        const int x0 = xc * kMxRowChunk, r0 = rb * 32;                 // r0: row of V
        // no pixel of the unit's own 128 is a mirror source: weights +-1 (uniform)
        const bool plain = x0 > g.pad && x0 + kMxRowChunk + g.pad + 1 < g.cols;
        float s[3];
        mx_row_commit<NKB, QUIRK>(raw, in, tid, g, x0, plain, s);
        if (QUIRK) {
            const int ri = r0 - PADA + (tid >> 3);                     // image row of the thread's row of V; mirrored rows are not counted
            if ((tid & 7) == 0 && ri >= 0 && ri < g.rows) {
                int* sp = spart + ((static_cast<size_t>(f) * chunks + xc) * g.rows + ri) * 3;
                sp[0] = static_cast<int>(s[0]); sp[1] = static_cast<int>(s[1]); sp[2] = static_cast<int>(s[2]);
            }
            if (tid < 32) {
                const int rw = r0 - PADA + tid;
                wyv[tid] = (rw >= 0 && rw < g.rows) ? static_cast<float>(mx_alt_weight(rw, g.rows, g.pad)) : 0.f;
            }
        }
        __syncthreads();
        // the next unit's pixels travel while this one is in the matrix cores
        if (u + ustep < uend) mx_row_issue<NKB>(raw, src, g, u + ustep, chunks, rblocks, tid);
        // ---- 4 tiles of 32 outputs x 3 channels = 12 products, 3 per wave
        mx_float16 acc[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int t = wave + 4 * tt, c = t % 3, tile = t / 3;
            const _Float16* base = in + (c * 32 + m) * PW + tile * 32 + 8 * h;
            mx_float16 a = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const mx_half8 x = *reinterpret_cast<const mx_half8*>(base + 16 * kb);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, th[kb], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, tl[kb], a, 0, 0, 0);
            }
            acc[tt] = a;
        }
        if (QUIRK) {
            // column sums of the tile: D[row][x] with the lane's 16 rows in registers -> 16 multiply-adds, one exchange with lane ^ 32
            float wy[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 t4 = *reinterpret_cast<const float4*>(wyv + 8 * k + 4 * h);
                wy[4 * k] = t4.x; wy[4 * k + 1] = t4.y; wy[4 * k + 2] = t4.z; wy[4 * k + 3] = t4.w;
            }
            float* vp = vpart + (static_cast<size_t>(f) * rblocks + rb) * g.vpitch;
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const int t = wave + 4 * tt, c = t % 3, tile = t / 3;
                float cs = 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) cs = __builtin_fmaf(acc[tt][reg], wy[reg], cs);
                cs += __shfl_xor(cs, 32);
                const int e = 3 * (x0 + tile * 32 + m) + c;
                if (h == 0 && e < g.vpitch) vp[e] = cs * kMxUnscale;
            }
        }
        __syncthreads();
        // ---- re-interleave through LDS: D[row][o], row = (reg & 3) + 8 (reg >> 2) + 4 h, o = lane & 31
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int t = wave + 4 * tt, c = t % 3, tile = t / 3;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                stage[row * kMxStagePitch + (tile * 32 + m) * 3 + c] = acc[tt][reg] * kMxUnscale;
            }
        }
        __syncthreads();
        {   // 12 strips x 4 row groups x 32 lanes = 1536 fragments of 24 bytes, six per thread; a wave stores 2 x 768 contiguous bytes
            unsigned char* vbase = reinterpret_cast<unsigned char*>(V) + static_cast<size_t>(f) * (g.vpitch / 32) * (g.vrows / 8) * 768;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int it = tid + 256 * k, seg = it >> 5, n = it & 31, sidx = seg >> 2, rgi = seg & 3;
                const int strip = (3 * x0) / 32 + sidx;
                if (strip < g.vpitch / 32) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = stage[(8 * rgi + j) * kMxStagePitch + 32 * sidx + n];
                    *reinterpret_cast<MxV24x8*>(vbase + (static_cast<size_t>(strip) * (g.vrows / 8) + (r0 / 8 + rgi)) * 768 + 24 * n) = mx_v24_pack(v);
                }
            }
        }
        __syncthreads();
    }
}

// v - float(hi) for both halves of a packed binary16 pair, one mixed-precision FMA each (hi * -1.0 + v)
__device__ __forceinline__ void mx_remainder(uint32_t hi2, float v0, float v1, float& r0, float& r1)
{
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi2), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi2), "v"(v1));
}

constexpr uint32_t kMxRsrcWord3 = 0x00020000u;   // raw buffer, 32-bit data format (gfx9 family)

// QUIRK: V[re][e] += qrow[f][c][re] * (-1)^x when a block is decoded (the row pass's Nyquist-slot term; qrow is indexed by the
// row of V, mirrored rows included), and out += qcol[f][e] * (-1)^r (the column pass's term); e = 3 x + c.  The qrow loads
// ride with the block's data loads -- same place in the in-order vmcnt queue, consumed together
// All global accesses are buffer instructions: resource = this wave's strip of this frame (output: this segment's rows of it),
// vector offset = the lane's constant byte offset, scalar offset = the block or row -- no vector arithmetic per access, and a
// store whose offset lies outside the resource (rows of another segment or past the image, columns past 3 cols) is dropped by
// its bounds check.
// Two waves per SIMD while 2 NKB fragments + NACC accumulator tiles fit 256 registers without spills (NKB = 11: the metric), then
// with 2 blocks in flight per wave; one wave per SIMD and 4 blocks otherwise.  Measured for NKB = 11 with the quirk's terms: 30.7
// against 32.1 us per frame.
#ifndef MX_COL_WAVES
#define MX_COL_WAVES(NKB_) ((NKB_) == 11 ? 2 : 1)
#endif
template <int NKB, bool QUIRK>
__global__ __launch_bounds__(256, MX_COL_WAVES(NKB)) void mx_colpass_u8(const float* __restrict__ V, uint8_t* __restrict__ dst, const mx_half8* __restrict__ frags, MxGeom g,
                                                        int nstrips, const float* __restrict__ qcol, int tps, int nseg, const float* __restrict__ qrow)
{
    // PD blocks in flight (PD divides the 2 NACC blocks of an unrolled round, so queue slots are compile-time); their loads, the
    // 16 byte stores of a tile and the quirk's term loads stay below the 63 the vmcnt counter can express
    constexpr int NACC = (NKB + 1) / 2, PD = MX_COL_WAVES(NKB) == 2 ? 2 : ((2 * NACC) % 4 == 0 ? 4 : ((2 * NACC) % 3 == 0 ? 3 : 2));
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    // task = (frame, strip, segment): few frames of a small image would leave most of the chip idle with one wave per strip,
    // so a strip is cut into `nseg` segments of `tps` output tiles (tps a multiple of NACC: the accumulator rotation below
    // starts aligned).  A segment begins with its own first window block (no run-in) and ends (NKB - 1) / 2 periods after
    // its last tile's first block: that tail is the only work done twice.
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int seg = task % nseg, fs = task / nseg;
    const int f = fs / nstrips, s = fs - f * nstrips;
    if (f >= g.nframes) return;
    const int e = 32 * s + n;
    const bool valid = e < 3 * g.cols;
    const uint32_t rowbytes = 3u * g.cols;
    const uint32_t stripbytes = static_cast<uint32_t>(g.vrows / 8) * 768u;
    const unsigned char* strip = reinterpret_cast<const unsigned char*>(V) + (static_cast<size_t>(f) * nstrips + s) * stripbytes;   // uniform
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(strip), 0, stripbytes, kMxRsrcWord3);
    uint8_t* ostrip = dst + static_cast<size_t>(f) * g.rows * rowbytes + 32 * s;                   // uniform
    const uint32_t lane_in = 24u * n + 768u * h;                                                   // byte offset inside a block (two row groups)
    const uint32_t lane_out = valid ? n + 4u * h * rowbytes : 0xfffffff0u;                         // invalid column: out of bounds

    mx_half8 th[NKB], tl[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        th[kb] = frags[kb * 64 + lane];
        tl[kb] = frags[(NKB + kb) * 64 + lane];
    }
    float cpos = 0.5f, cneg = 0.5f;                      // + 0.5f of the u8 conversion, +- the column term on even / odd rows
    float sgn_x = 0.f;
    __amdgpu_buffer_rsrc_t rq = rin;
    uint32_t lane_q = 0;
    if (QUIRK) {
        const float qc = valid ? qcol[static_cast<size_t>(f) * g.vpitch + e] : 0.f;
        cpos = 0.5f + qc;
        cneg = 0.5f - qc;
        sgn_x = valid ? (((e / 3) & 1) ? -1.f : 1.f) : 0.f;
        rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qrow) + static_cast<size_t>(f) * 3 * g.vrows, 0, 12u * g.vrows, kMxRsrcWord3);
        lane_q = 4u * (static_cast<uint32_t>(e % 3) * g.vrows + 8u * h);                          // the lane's channel, its 8 rows of a block
    }

    const int nblocks = g.vrows / 16;
    constexpr int QW = 6;
    auto load_block = [&](int jb, uint32_t (&d)[6]) {
        // blocks past the end are never part of an emitted tile: read the last one again instead
        const uint32_t off = 1536u * min(jb, nblocks - 1);                                          // uniform
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rin, lane_in, off, 0);
        const u2 b = __builtin_amdgcn_raw_buffer_load_b64(rin, lane_in + 16u, off, 0);
        d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3]; d[4] = b[0]; d[5] = b[1];
    };
    auto load_terms = [&](int jb, float (&q)[8]) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const uint32_t off = 64u * min(jb, nblocks - 1);                                            // uniform: 16 rows x 4 bytes
        const f4 a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rq, lane_q, off, 0));
        const f4 b = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rq, lane_q + 16u, off, 0));
        q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = a[3]; q[4] = b[0]; q[5] = b[1]; q[6] = b[2]; q[7] = b[3];
    };
    uint32_t queue[PD][QW];
    float qqueue[QUIRK ? PD : 1][8];
    const int ntiles = (g.rows + 31) / 32;
    const int tile0 = seg * tps, tile1 = min(tile0 + tps, ntiles);             // this segment's output tiles
    // the output resource covers exactly the segment's rows of the strip: rows of other segments (the incomplete tiles of the
    // run-in, the repeated ones of the run-out), rows past the image and columns past 3 cols all fail its bounds check
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(ostrip + static_cast<size_t>(32 * tile0) * rowbytes, 0,
                                                                          static_cast<uint32_t>(min(32 * tile1, g.rows) - 32 * tile0) * rowbytes - 32u * s, kMxRsrcWord3);
#pragma unroll
    for (int k = 0; k < PD; ++k) {
        load_block(2 * tile0 + k, queue[k]);
        if (QUIRK) load_terms(2 * tile0 + k, qqueue[k]);
    }

    mx_float16 acc[NACC];
    const mx_float16 zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = zero;
    const int pend = tile1 + (NKB - 1) / 2;

    for (int mp = tile0; mp < pend; mp += NACC) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            const int p = mp + q;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int slotq = (2 * q + b) % PD;
                // split the block into hi + lo halves (round to nearest even both times: the remainders have zero mean)
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                u4 w1, w2;
                float vv[8];
                mx_v24_unpack(queue[slotq], vv);
                if (QUIRK) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) vv[k] = __builtin_fmaf(qqueue[slotq][k], sgn_x, vv[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f2 v = { vv[2 * k], vv[2 * k + 1] };
                    const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
                    float r0, r1;
                    mx_remainder(hi, v[0], v[1], r0, r1);
                    const f2 rem = { r0, r1 };
                    w1[k] = hi;
                    w2[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, h2));
                }
                const mx_half8 v1 = __builtin_bit_cast(mx_half8, w1), v2 = __builtin_bit_cast(mx_half8, w2);
                load_block(2 * p + b + PD, queue[slotq]);
                if (QUIRK) load_terms(2 * p + b + PD, qqueue[slotq]);
#pragma unroll
                for (int a = 0; 2 * a + b < NKB; ++a) {
                    const int d = 2 * a + b, slot = (q - a + 2 * NACC) % NACC;
                    mx_float16 c = d == 0 ? zero : acc[slot];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[d], v1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[d], v1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[d], v2, c, 0, 0, 0);
                    acc[slot] = c;
                }
                if ((NKB - 1 - b) % 2 == 0) {            // the tile whose last window block this was
                    // (tiles before the first and past the last are stored too: their rows lie outside the resource and the
                    // bounds check drops them -- no branch, so the whole period stays one scheduling region)
                    const int a = (NKB - 1 - b) / 2, slot = (q - a + 2 * NACC) % NACC, tile = p - a;
                    {
                        // uniform, relative to the segment's first row: tiles before the segment wrap far out of bounds
                        const uint32_t orow0 = 32u * static_cast<uint32_t>(tile - tile0) * rowbytes;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const float v = __builtin_fmaf(acc[slot][reg], kMxUnscale, (reg & 1) ? cneg : cpos);
                            __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(static_cast<int>(v)), rout, lane_out,
                                                                 orow0 + static_cast<uint32_t>((reg & 3) + 8 * (reg >> 2)) * rowbytes, 0);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The Nyquist-slot quirk of pffft_() (Source.cpp:420-425) without an FFT.  The reference scales bin N/2 of every padded
// line with the DC gain; the difference to the true multiplier, delta = m[0] - m[N/2], reaches the output as
//     delta * (-1)^n * sum_p (-1)^p line[p]            (n, p: positions in the padded line; the zero tail adds nothing)
// i.e. one alternating sum per line.  With reflect-101 borders that sum is a weighted sum of the image line itself:
// pixel i sits at p = i + pad and its mirror images (1 <= i <= pad on the left, len-1-pad <= i <= len-2 on the right)
// all land on positions of the same parity, so weight(i) = (-1)^(i+pad) * (1 + [left mirror] + [right mirror]).
// Both passes together (wx, wy: the weights along a row and along a column):
//     V'[r][x]  = V[r][x] + dr (-1)^(x+pad) Srow(r),          Srow(r) = sum_x wx(x) img[r][x]                    (integers)
//     out[r][x] = colconv(V')[r][x] + dc (-1)^(r+pad) Scol(x),  Scol(x) = sum_r wy(r) V'[r][x]
//                                                                     = sum_r wy(r) V[r][x] + dr (-1)^(x+pad) Z,  Z = sum_r wy(r) Srow(r)
// The row kernel leaves the partial sums behind (spart: Srow per 128-pixel chunk; vpart: the wy-weighted column sums of V per
// 32-row block); mx_quirk_rows and mx_quirk_cols add them up in a fixed order and produce the two float vectors the column
// kernel adds: qrow (per row of V, when a block is decoded) and qcol (per column, when a tile leaves).
#ifdef BLUR_MX_QUIRK_KERNELS   // engine.hip only: plain (non-template) kernels must live in one translation unit
// Rows: Srow[r][c] = sum over the chunks of spart (integers); qrow[f][c][re] = dr (-1)^pad Srow[refl(re - PADA)][c] for every
// row re of V; zpart[f][block][c] = the block's part of Z_c = sum_r wy(r) Srow[r][c].  grid: (ceil(vrows / 256), frames)
__global__ __launch_bounds__(256) void mx_quirk_rows(const int* __restrict__ spart, float* __restrict__ qrow, double* __restrict__ zpart, MxGeom g, int chunks,
                                                     int pada, float dr)
{
    __shared__ double zs[3][256];
    const int f = blockIdx.y, tid = threadIdx.x, re = blockIdx.x * 256 + tid;
    const double sp = (g.pad & 1) ? -1.0 : 1.0;
    double z[3] = { 0, 0, 0 };
    if (re < g.vrows) {
        const int r = mx_refl(re - pada, g.rows);
        const bool own = re - pada >= 0 && re - pada < g.rows;           // every image row is counted once in Z
        const int* p = spart + (static_cast<size_t>(f) * chunks * g.rows + r) * 3;
        int sum[3] = { 0, 0, 0 };
        for (int k = 0; k < chunks; ++k) {
            const int* q = p + static_cast<size_t>(k) * g.rows * 3;
            sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2];
        }
        const double wy = own ? static_cast<double>(mx_alt_weight(r, g.rows, g.pad)) : 0.0;
        for (int c = 0; c < 3; ++c) {
            qrow[(static_cast<size_t>(f) * 3 + c) * g.vrows + re] = static_cast<float>(static_cast<double>(dr) * sp * sum[c]);
            z[c] = wy * sum[c];
        }
    }
    for (int c = 0; c < 3; ++c) zs[c][tid] = z[c];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) for (int c = 0; c < 3; ++c) zs[c][tid] += zs[c][tid + o];
        __syncthreads();
    }
    if (tid < 3) zpart[(static_cast<size_t>(f) * gridDim.x + blockIdx.x) * 3 + tid] = zs[tid][0];
}

// Columns: Scol[e] = sum over the row blocks of vpart + dr (-1)^(x+pad) Z_c; qcol[f][e] = dc (-1)^pad Scol[e].
// grid: (ceil(vpitch / 256), frames)
__global__ __launch_bounds__(256) void mx_quirk_cols(const float* __restrict__ vpart, const double* __restrict__ zpart, float* __restrict__ qcol, MxGeom g,
                                                     int rblocks, int zblocks, float dr, float dc)
{
    const int f = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= g.vpitch) return;
    const int x = e / 3, c = e - 3 * x;
    double zc = 0;
    for (int k = 0; k < zblocks; ++k) zc += zpart[(static_cast<size_t>(f) * zblocks + k) * 3 + c];
    const float* p = vpart + static_cast<size_t>(f) * rblocks * g.vpitch + e;
    double acc = 0;
    int k = 0;
    for (; k + 8 <= rblocks; k += 8) {                       // eight independent loads per trip
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = p[static_cast<size_t>(k + j) * g.vpitch];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += t[j];
    }
    for (; k < rblocks; ++k) acc += p[static_cast<size_t>(k) * g.vpitch];
    const double sp = (g.pad & 1) ? -1.0 : 1.0, sx = ((x + g.pad) & 1) ? -1.0 : 1.0;
    qcol[static_cast<size_t>(f) * g.vpitch + e] = x < g.cols ? static_cast<float>(static_cast<double>(dc) * sp * (acc + static_cast<double>(dr) * sx * zc)) : 0.f;
}
#endif  // BLUR_MX_QUIRK_KERNELS

// ---- launchers: one translation unit per NKB (mx_conv_<NKB>.hip) --------------------------------------------------
struct MxEntry {
    int nkb;          // window blocks: pad <= 8 (nkb - 2)
    hipError_t (*row_u8)(hipStream_t, const uint8_t* src, float* V, const void* frags, MxGeom g, int num_cus, int* spart, float* vpart);
    hipError_t (*col_u8)(hipStream_t, const float* V, uint8_t* dst, const void* frags, MxGeom g, const float* qcol, int num_cus, const float* qrow);
};

template <int NKB> hipError_t mx_launch_row_u8(hipStream_t st, const uint8_t* src, float* V, const void* frags, MxGeom g, int num_cus, int* spart, float* vpart)
{
    const int chunks = (g.cols + kMxRowChunk - 1) / kMxRowChunk, rblocks = g.vrows / 32;
    const long long nunits = static_cast<long long>(chunks) * rblocks * g.nframes;
    if (nunits <= 0) return hipSuccess;
    const size_t lds = mx_row_lds(NKB) + 32 * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mx_rowpass_u8<NKB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(mx_rowpass_u8<NKB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    const int per_cu = (lds <= 80 * 1024 && NKB <= 17) ? 2 : 1;
    const int grid = static_cast<int>(nunits < static_cast<long long>(num_cus) * per_cu ? nunits : static_cast<long long>(num_cus) * per_cu);
    if (spart)
        hipLaunchKernelGGL((mx_rowpass_u8<NKB, true>), dim3(grid), dim3(256), lds, st, src, V, static_cast<const mx_half8*>(frags), g, chunks, rblocks,
                           static_cast<int>(nunits), spart, vpart);
    else
        hipLaunchKernelGGL((mx_rowpass_u8<NKB, false>), dim3(grid), dim3(256), lds, st, src, V, static_cast<const mx_half8*>(frags), g, chunks, rblocks,
                           static_cast<int>(nunits), spart, vpart);
    return hipGetLastError();
}

template <int NKB> hipError_t mx_launch_col_u8(hipStream_t st, const float* V, uint8_t* dst, const void* frags, MxGeom g, const float* qcol, int num_cus,
                                               const float* qrow)
{
    constexpr int NACC = (NKB + 1) / 2;
    const int nstrips = g.vpitch / 32, ntiles = (g.rows + 31) / 32;
    const long long strips = static_cast<long long>(nstrips) * g.nframes;
    if (strips <= 0) return hipSuccess;
    // segments per strip: as few as fill the chip's 4 waves per CU about once (every segment repeats (NKB - 1) / 2 periods)
    int tps = ((ntiles + NACC - 1) / NACC) * NACC;
    while (tps > NACC && strips * ((ntiles + tps - 1) / tps) < 4ll * num_cus) tps -= NACC;
    const int nseg = (ntiles + tps - 1) / tps;
    const long long tasks = strips * nseg;
    const dim3 grid(static_cast<unsigned>((tasks + 3) / 4));
    if (qcol) hipLaunchKernelGGL((mx_colpass_u8<NKB, true>), grid, dim3(256), 0, st, V, dst, static_cast<const mx_half8*>(frags), g, nstrips, qcol, tps, nseg, qrow);
    else hipLaunchKernelGGL((mx_colpass_u8<NKB, false>), grid, dim3(256), 0, st, V, dst, static_cast<const mx_half8*>(frags), g, nstrips, qcol, tps, nseg, qrow);
    return hipGetLastError();
}

#define BLUR_MX(NKB_)                                                                                       \
    namespace blur_amd {                                                                                    \
    const MxEntry* mx_entry_##NKB_()                                                                        \
    {                                                                                                       \
        static const MxEntry e = { NKB_, mx_launch_row_u8<NKB_>, mx_launch_col_u8<NKB_> };                  \
        return &e;                                                                                          \
    }                                                                                                       \
    }

}  // namespace blur_amd
