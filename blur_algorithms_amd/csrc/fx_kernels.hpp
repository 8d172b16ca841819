// fx_kernels.hpp -- the separable blur as ONE kernel on the f16 matrix cores (gfx950): row pass and column pass fused, the
// intermediate V never leaves the register file.
//
// The two-kernel matrix-core engine (mx_kernels.hpp) moves 25 B/px through HBM for 6 B/px of image and spends most of its time
// on the 19 B/px of V.  Here a workgroup of four waves (one per SIMD, 512 registers each) owns a strip of 128 pixel columns
// and marches down the image in steps of 32 rows:
//
//   stage     the step's window (32 rows x (128 + 2 PADA) pixels x 3 channels) arrives as 12-byte groups one step ahead
//             (registers), is deinterleaved + converted to binary16 with v_perm_b32 and parked in LDS (double buffered:
//             one barrier per step)
//   row pass  wave w owns pixels 32 w .. 32 w + 31 of the strip, all three channels: D[32 rows][32 pixels] =
//             window (A, ds_read_b128) x Toeplitz fragment (B, registers), taps split hi + lo: 2 NKB products per channel
//   hand-off  D is exactly the column pass's B operand up to an exchange between lane l and lane l ^ 32: scale (+ the
//             quirk's row term), split into hi + lo binary16, two v_permlane32_swap per operand and 16-row block
//   col pass  "sliding accumulators" as in mx_colpass_u8: each 16-row block of V is multiplied into the (NKB + 1) / 2
//             output tiles whose windows contain it (A = the same Toeplitz fragments), v_hi t_hi + v_hi t_lo + v_lo t_hi;
//             (NKB - 1) / 2 accumulator tiles per channel stay live (the tile that finishes in a step hands its registers
//             to the tile that starts in it), the rotation is static: the step loop is unrolled (NKB - 1) / 2 times
//   emit      a finished tile (32 rows x 32 pixels x 3 channels) is scaled, + 0.5f-truncated, packed to bytes, transposed
//             inside lane quads (DPP + v_perm_b32) so that a lane owns 4 adjacent pixels = 12 contiguous bytes of one
//             image row, and stored with global_store_dwordx3: a wave store = 8 rows x 96 contiguous bytes
//
// HBM traffic: 3 B/px read (the window overlap of neighbouring strips is served by L2: strips of one frame run on one XCD)
// + 3 B/px written.  The kernel is bound by the matrix pipe: 55 MFMAs per (32 x 32 tile, channel) for NKB = 11.
//
// The Nyquist-slot quirk of pffft_() (Source.cpp:420-425; notes in mx_kernels.hpp) needs the alternating sums of whole image
// rows and columns BEFORE a pixel can leave, so they come from a pre-pass over the image (fx_altsums: exact integers) and two
// small kernels that turn them into the per-row and per-column terms qrow / qcol; the column sums of V the column term is
// made of are, by linearity, the row convolution of the image's weighted column sums.
#pragma once
#include "mx_kernels.hpp"

#ifndef FX_SGB_A
#define FX_SGB_A 6
#endif
#ifndef FX_SGB_B
#define FX_SGB_B 5
#endif

namespace blur_amd {

struct FxGeom {
    int rows, cols, pad;
    int nframes;
    int aligned;      // 1: cols % 4 == 0 and frame pointers 4-byte aligned: 12-byte pixel groups are three aligned dwords
    int ntiles;       // output tiles of 32 rows per frame
};

constexpr int kFxChunk = 128;     // pixel columns per workgroup

template <int NKB> struct FxCfg {
    static constexpr int PADA = 8 * (NKB - 2), WIN = kFxChunk + 2 * PADA, GPR = WIN / 4, PER = (GPR + 7) / 8;
    static constexpr int PW = mx_row_pitch(NKB);                     // halfs per LDS row of the window
    static constexpr int NT = (NKB - 1) / 2;                          // live accumulator tiles per channel = steps per unrolled round
    static constexpr int BUF = 3 * 32 * PW * 2;                       // bytes of one window buffer
    static constexpr int QOFF = 2 * BUF;                              // qrow stage: [2][3][32] floats
    static constexpr int LDS = QOFF + 2 * 96 * 4;
};

// x as binary16: the byte in the low half of a binary16 is the SUBNORMAL x * 2^-24 -- the matrix cores take subnormal
// operands at full precision, so the conversion is the deinterleaving v_perm_b32 itself (one instruction per two values)
constexpr float kFxRowUnscale = 1024.f;          // V = acc * 2^24 / 2^14

// selector byte of v_perm_b32 for byte `B` (0..11) of a 12-byte group held in dwords d[0..2], given which dwords sit in
// (a = high, b = low) operand slots
__host__ __device__ constexpr uint32_t fx_sel1(int B, int da, int db) { return (B >> 2) == db ? static_cast<uint32_t>(B & 3) : static_cast<uint32_t>(4 + (B & 3)); }

template <int NKB> struct FxRaw { uint32_t d[FxCfg<NKB>::PER][3]; };

// upper half of a <-> lower half of b (lanes l and l ^ 32), four pairs per statement; see wr_kernels.hpp: wr_swap4 for the
// wait states
__device__ __forceinline__ void fx_swap4(uint32_t& a0, uint32_t& b0, uint32_t& a1, uint32_t& b1, uint32_t& a2, uint32_t& b2, uint32_t& a3, uint32_t& b3)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
        : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
}

template <int CTRL> __device__ __forceinline__ uint32_t fx_dpp(uint32_t x)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xf, 0xf, true));
}

// 4 x 4 byte transpose inside a lane quad: lane j holds bytes M[j][0..3]; afterwards lane k holds M[0..3][k]
__device__ __forceinline__ uint32_t fx_quad_transpose(uint32_t p, uint32_t sel1, uint32_t sel2)
{
    const uint32_t t = fx_dpp<0xb1>(p);                       // quad_perm [1,0,3,2]
    const uint32_t q = __builtin_amdgcn_perm(t, p, sel1);
    const uint32_t u = fx_dpp<0x4e>(q);                       // quad_perm [2,3,0,1]
    return __builtin_amdgcn_perm(u, q, sel2);
}

// One workgroup per (frame, segment of output tiles, chunk of 128 pixel columns).
//   qrow[f][c][re]   the row pass's quirk term per row of V (re = image row + PADA, mirrored rows included), added as qrow * (-1)^x
//   qcol[f][3 x + c] the column pass's quirk term, added as qcol * (-1)^r
//
// Software pipeline.  n = (step s, channel c) enumerates the wave's (32 rows x 32 pixels) products; per n the matrix pipe runs
//     phase A(n):  R(n+1)  row pass of the NEXT product (2 NKB MFMAs)            beside: staging of later windows, the stores
//     phase B(n):  C(n)    column pass of this one (3 NKB MFMAs)                 beside: hand-off S(n+1), emission E(n)
// so every vector instruction has matrix work of another product to hide behind, and exactly 16 accumulator tiles are live
// (15 of the column pass + the row pass's): C(n) begins with the tile that FINISHES (its last window block) and ends with the
// tile that STARTS, which takes over the finished tile's registers after E(n) has read them; the row accumulator is read out
// by S(n+1) early in B(n), before R(n+2) writes it again.
template <int NKB, bool QUIRK>
__global__ __launch_bounds__(256, 1) void fx_blur_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const mx_half8* __restrict__ frags, FxGeom g,
                                                     int chunks, int tps, int nseg, int ntasks, const float* __restrict__ qrow, const float* __restrict__ qcol,
                                                     int qpitch)
{
    using C = FxCfg<NKB>;
    constexpr int PADA = C::PADA, PW = C::PW, NT = C::NT, PER = C::PER;
    static_assert(PER <= 9, "three staging chunks of three 12-byte groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char fx_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;

    // tasks are numbered chunk-fastest; workgroups b and b + 8 share an XCD: every XCD gets a contiguous band of tasks, so the
    // strips left and right of a strip -- which read 2 PADA of its window columns -- run on the same L2 at about the same time
    const int xcd = blockIdx.x & 7, per_xcd = (ntasks + 7) / 8, task = xcd * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || task >= ntasks) return;
    const int xc = task % chunks, seg = (task / chunks) % nseg, f = task / (chunks * nseg);
    const int x0 = xc * kFxChunk;
    const int tile0 = seg * tps, tile1 = min(tile0 + tps, g.ntiles);
    const uint8_t* img = src + static_cast<size_t>(f) * g.rows * g.cols * 3;
    uint8_t* out = dst + static_cast<size_t>(f) * g.rows * g.cols * 3;

    // hi halves of the fragments in registers, lo halves in LDS (one ds_read_b128 per use: the register file holds the
    // 15 accumulator tiles instead)
    // the Toeplitz fragments (hi and lo halves) serve both passes: B operand of the row pass, A operand of the column pass
    mx_half8 th[NKB], tl[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        th[kb] = frags[kb * 64 + lane];
        tl[kb] = frags[(NKB + kb) * 64 + lane];
    }

    // per-lane constants of the emission: quad transposes and the store address
    const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
    const int Q = m >> 2, q = m & 3;
    const int xpix = x0 + 32 * wave + 4 * Q;                       // first of the lane's 4 pixels after the transposes
    const bool in_cols = xpix < g.cols;
    float cpos[3], cneg[3], sgnx = 0.f;
    {
        const int x = x0 + 32 * wave + m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float qc = 0.f;
            if (QUIRK && x < g.cols) qc = qcol[static_cast<size_t>(f) * qpitch + 3 * x + c];
            cpos[c] = 0.5f + qc;
            cneg[c] = 0.5f - qc;
        }
        sgnx = (x & 1) ? -1.f : 1.f;
    }
    const int qrows = 32 * (g.ntiles + NT);

    const mx_float16 zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    mx_float16 acc[3][NT];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[c][k] = zero;
    mx_float16 arow = zero;             // the row pass's accumulator
    uint32_t hl[2][2][8];               // hand-off: [buffer][hi, lo][packed row pairs], block b = entries 4 b .. 4 b + 3
    uint32_t rr[3][4];                  // finished tile, per channel and row group: 4 pixels of one row (after the quad transpose)

    // step s handles rows 32 s .. 32 s + 31 of V (row re of V = image row refl(re - PADA)) and emits tile s - NT
    const int s0 = tile0, s1 = tile1 + NT;
    FxRaw<NKB> raw;
    float qraw = 0.f;
    // staging in three chunks of three 12-byte groups: commit chunk j of window s, then refill the registers with window s + 1
    // Reflect-101 along the rows (Source.cpp:525-529) is done by the loads: the image width is a multiple of 4, so a 12-byte group
    // of the window lies inside the image or outside it, and a group outside is the pixel-reversed copy of a group inside, 3 bytes
    // (left) or 1 byte (right) off the dword grid -- gfx950 loads it unaligned; the commit reverses its pixels (other selectors of
    // the same v_perm_b32).  Only the chunks at the two edges (uniform per workgroup) run that code.  The frame is a buffer
    // resource: mirrored groups beyond the reach of the taps (tiny images) may fall outside it and read as zero.
    const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, static_cast<uint32_t>(g.rows) * g.cols * 3u, kMxRsrcWord3);
    const bool edge = x0 - PADA < 0 || x0 + kFxChunk + PADA > g.cols;      // uniform
    auto group_x = [&](int k, bool& mir) __attribute__((always_inline)) {  // first source pixel of group k of this thread
        const int X = x0 - PADA + 4 * ((tid & 7) + 8 * k);
        mir = X < 0 || X >= g.cols;
        return X < 0 ? -X - 3 : (X >= g.cols ? 2 * g.cols - 5 - X : X);
    };
    auto issue_chunk = [&](int s, int j) __attribute__((always_inline)) {
        const int row = tid >> 3, g0 = tid & 7;
        const int r = mx_refl(32 * s - PADA + row, g.rows);
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
        if (!edge) {
            const uint32_t off = (static_cast<uint32_t>(r) * g.cols + static_cast<uint32_t>(x0 - PADA + 4 * g0)) * 3u;
#pragma unroll
            for (int k = 3 * j; k < 3 * j + 3 && k < PER; ++k) {
                const bool in = (C::GPR % 8 == 0) || k < PER - 1 || g0 < C::GPR % 8;
                const u3 t = __builtin_amdgcn_raw_buffer_load_b96(rimg, in ? off + 96u * k : off, 0, 0);
                raw.d[k][0] = t[0]; raw.d[k][1] = t[1]; raw.d[k][2] = t[2];
            }
        } else {
            const uint32_t rowoff = static_cast<uint32_t>(r) * g.cols * 3u;
#pragma unroll
            for (int k = 3 * j; k < 3 * j + 3 && k < PER; ++k) {
                bool mir;
                const int xs = group_x(k, mir);
                const u3 t = __builtin_amdgcn_raw_buffer_load_b96(rimg, rowoff + 3u * static_cast<uint32_t>(xs), 0, 0);
                raw.d[k][0] = t[0]; raw.d[k][1] = t[1]; raw.d[k][2] = t[2];
            }
        }
        if (QUIRK && j == 0 && tid < 96) {
            const int c = tid >> 5, re = 32 * s + (tid & 31);
            qraw = qrow[(static_cast<size_t>(f) * 3 + c) * qrows + min(re, qrows - 1)];
        }
    };
    auto commit_chunk = [&](int buf, int j) __attribute__((always_inline)) {
        const int row = tid >> 3, g0 = tid & 7;
        _Float16* base = reinterpret_cast<_Float16*>(fx_lds + buf * C::BUF) + row * PW + 4 * g0;
#pragma unroll
        for (int k = 3 * j; k < 3 * j + 3 && k < PER; ++k) {
            if ((C::GPR % 8 == 0) || k < PER - 1 || g0 < C::GPR % 8) {
                bool mir = false;
                if (edge) (void)group_x(k, mir);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // window pixels (0, 1) and (2, 3) of the group: source bytes (c, 3 + c) and (6 + c, 9 + c); mirrored: (9 + c, 6 + c) and (3 + c, c)
                    auto pick = [&](int B0, int B1) __attribute__((always_inline)) {
                        const int da = B1 >> 2, db = B0 >> 2;
                        const uint32_t sel = fx_sel1(B0, da, db) | (0x0cu << 8) | (fx_sel1(B1, da, db) << 16) | (0x0cu << 24);
                        return __builtin_amdgcn_perm(raw.d[k][da], raw.d[k][db], sel);
                    };
                    uint2 wd;
                    wd.x = pick(c, 3 + c);
                    wd.y = pick(6 + c, 9 + c);
                    if (edge) {
                        const uint32_t mx_ = pick(9 + c, 6 + c), my_ = pick(3 + c, c);
                        wd.x = mir ? mx_ : wd.x;
                        wd.y = mir ? my_ : wd.y;
                    }
                    *reinterpret_cast<uint2*>(base + c * 32 * PW + 32 * k) = wd;
                }
            }
        }
        if (QUIRK && j == 0 && tid < 96) reinterpret_cast<float*>(fx_lds + C::QOFF)[buf * 96 + tid] = qraw;
    };
    // R: 32 rows x 32 pixels of channel c of the window in buffer `buf` -> arow
    auto rowpass = [&](int buf, int c) __attribute__((always_inline)) {
        const _Float16* base = reinterpret_cast<const _Float16*>(fx_lds + buf * C::BUF) + (c * 32 + m) * PW + wave * 32 + 8 * h;
        mx_float16 a = zero;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const mx_half8 x = *reinterpret_cast<const mx_half8*>(base + 16 * kb);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, th[kb], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, tl[kb], a, 0, 0, 0);
        }
        arow = a;
    };
    // S: arow -> scale (+ quirk), split into hi + lo, exchange with lane ^ 32 -> hl[hb]
    auto split = [&](int hb, int buf, int c) __attribute__((always_inline)) {
        float v[16];
        if (QUIRK) {
            const float* qs4 = reinterpret_cast<const float*>(fx_lds + C::QOFF) + buf * 96 + c * 32 + 4 * h;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 t4 = *reinterpret_cast<const float4*>(qs4 + 8 * k);
                v[4 * k] = __builtin_fmaf(arow[4 * k], kFxRowUnscale, t4.x * sgnx);
                v[4 * k + 1] = __builtin_fmaf(arow[4 * k + 1], kFxRowUnscale, t4.y * sgnx);
                v[4 * k + 2] = __builtin_fmaf(arow[4 * k + 2], kFxRowUnscale, t4.z * sgnx);
                v[4 * k + 3] = __builtin_fmaf(arow[4 * k + 3], kFxRowUnscale, t4.w * sgnx);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = arow[k] * kFxRowUnscale;
        }
        uint32_t (&hp)[8] = hl[hb][0];
        uint32_t (&lp)[8] = hl[hb][1];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const f2 vv = { v[2 * k], v[2 * k + 1] };
            hp[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(vv, h2));
            float r0, r1;
            mx_remainder(hp[k], vv[0], vv[1], r0, r1);
            const f2 rem = { r0, r1 };
            lp[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, h2));
        }
        // regs (0..3, 4..7) = rows (0..3, 8..11) + 4 h -> block 0 wants rows 8 h .. 8 h + 7; same for regs 8..15 and block 1
        fx_swap4(hp[0], hp[2], hp[1], hp[3], lp[0], lp[2], lp[1], lp[3]);
        fx_swap4(hp[4], hp[6], hp[5], hp[7], lp[4], lp[6], lp[5], lp[7]);
    };
    // C + E: column pass of (step slot qs, channel c) from hl[hb]; the finished tile's bytes -> rr[c]
    auto colpass = [&](int hb, int c, int qs) __attribute__((always_inline)) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        mx_half8 v1[2], v2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const u4 w1 = { hl[hb][0][4 * b], hl[hb][0][4 * b + 1], hl[hb][0][4 * b + 2], hl[hb][0][4 * b + 3] };
            const u4 w2 = { hl[hb][1][4 * b], hl[hb][1][4 * b + 1], hl[hb][1][4 * b + 2], hl[hb][1][4 * b + 3] };
            v1[b] = __builtin_bit_cast(mx_half8, w1);
            v2[b] = __builtin_bit_cast(mx_half8, w2);
        }
        {   // the tile that finishes with block 0 of this step (window block NKB - 1)
            constexpr int d = NKB - 1;
            const int slot = qs % NT;
            mx_float16 t = acc[c][slot];
            const mx_half8 fh = th[d], fl = tl[d];
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, v1[0], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, v1[0], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, v2[0], t, 0, 0, 0);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint32_t by[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int reg = 4 * gq + k;
                    by[k] = static_cast<uint32_t>(static_cast<int>(__builtin_fmaf(t[reg], kMxUnscale, (reg & 1) ? cneg[c] : cpos[c])));
                }
                const uint32_t t01 = __builtin_amdgcn_perm(by[1], by[0], 0x0c0c0400u), t23 = __builtin_amdgcn_perm(by[3], by[2], 0x04000c0cu);
                rr[c][gq] = fx_quad_transpose(t01 | t23, sel1, sel2);
            }
        }
        // window blocks 2 .. NKB - 2 of the tiles in flight, then blocks 0 and 1 of the tile that starts (into the finished tile's
        // registers)
#pragma unroll
        for (int dd = 2; dd < NKB + 1; ++dd) {
            const int d = dd >= NKB - 1 ? dd - (NKB - 1) : dd, b = d & 1, a2 = d >> 1, slot = (qs - a2 + 2 * NT) % NT;
            mx_float16 t = d == 0 ? zero : acc[c][slot];
            const mx_half8 fh = th[d], fl = tl[d];
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, v1[b], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, v1[b], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, v2[b], t, 0, 0, 0);
            acc[c][slot] = t;
        }
    };
    // F: the finished tile (rr) -> 12 interleaved bytes per lane and row group -> memory
    auto store_tile = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const uint32_t r0 = rr[0][gq], r1 = rr[1][gq], r2 = rr[2][gq];
            // lane q of the quad holds row q: r_c = channel c of pixels 0..3 -> 12 interleaved bytes
            const uint32_t X = __builtin_amdgcn_perm(r1, r0, 0x05010400u);      // [r0.0, r1.0, r0.1, r1.1]
            const uint32_t Y = __builtin_amdgcn_perm(r1, r0, 0x07030602u);      // [r0.2, r1.2, r0.3, r1.3]
            const uint32_t w0 = __builtin_amdgcn_perm(r2, X, 0x02040100u);      // [X0, X1, r2.0, X2]
            const uint32_t Z = __builtin_amdgcn_perm(r2, X, 0x0c0c0503u);       // [X3, r2.1, 0, 0]
            const uint32_t w1 = __builtin_amdgcn_perm(Y, Z, 0x05040100u);       // [Z0, Z1, Y0, Y1]
            const uint32_t w2 = __builtin_amdgcn_perm(r2, Y, 0x07030206u);      // [r2.2, Y2, Y3, r2.3]
            const int row = 32 * tile + 8 * gq + 4 * h + q;
            if (row < g.rows && in_cols) {                              // (cols is a multiple of 4: a quad is inside or outside)
                typedef uint32_t u3 __attribute__((ext_vector_type(3)));
                const u3 w = { w0, w1, w2 };
                *reinterpret_cast<u3*>(out + (static_cast<size_t>(row) * g.cols + xpix) * 3) = w;
            }
        }
    };

    // prologue: window s0 and chunk 0 of window s0 + 1 in LDS; in the registers chunks 1, 2 of window s0 + 1 and chunk 0 of
    // window s0 + 2, as the loop expects them; R and S of the first product done
#pragma unroll
    for (int j = 0; j < 3; ++j) issue_chunk(s0, j);
#pragma unroll
    for (int j = 0; j < 3; ++j) commit_chunk(0, j);
    issue_chunk(s0 + 1, 0);
    commit_chunk(1, 0);
    issue_chunk(s0 + 1, 1);
    issue_chunk(s0 + 1, 2);
    issue_chunk(s0 + 2, 0);
    __syncthreads();
    rowpass(0, 0);
    split(0, 0, 0);

    for (int sb = s0; sb < s1; sb += NT) {
#pragma unroll
        for (int qs = 0; qs < NT; ++qs) {
            const int s = sb + qs;
            if (s >= s1) break;
            const int cur = (s - s0) & 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int hb = (3 * qs + c) & 1;                           // hand-off buffer (static: see the copy after the round)
                // ---- phase A: the next product's row pass; staging of window s + 2 (chunks in A(s,2), A(s+1,0), A(s+1,1)); stores
                if (c == 2) __syncthreads();                               // window s + 1 complete, window s no longer read
#ifndef FX_NOSB
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (c == 2) rowpass(cur ^ 1, 0); else rowpass(cur, c + 1);
                if (c == 2) { commit_chunk(cur, 0); issue_chunk(s + 3, 0); }
                else if (c == 0) { commit_chunk(cur ^ 1, 1); issue_chunk(s + 2, 1); }
                else { commit_chunk(cur ^ 1, 2); issue_chunk(s + 2, 2); }
                if (c == 0 && s - 1 - NT >= tile0 && s > s0) store_tile(s - 1 - NT);
#ifdef FX_SGB
                // the order the scheduler is asked for: window reads three blocks ahead, staging spread between the products
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int i = 0; i < NKB; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_A, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#endif
#ifndef FX_NOSB
                __builtin_amdgcn_sched_barrier(0);
#endif
                // ---- phase B: this product's column pass; the next product's hand-off; this tile's bytes
                colpass(hb, c, qs);
                if (c == 2) split(hb ^ 1, cur ^ 1, 0); else split(hb ^ 1, cur, c + 1);
#ifdef FX_SGB
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
                for (int i = 0; i < 3 * NKB - 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_B, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
#endif
#ifndef FX_NOSB
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        // a round is 3 NT products, an odd number: its last hand-off went to buffer 1, the next round starts reading buffer 0
        if ((3 * NT) & 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { hl[0][0][k] = hl[1][0][k]; hl[0][1][k] = hl[1][1][k]; }
        }
    }
    if (s1 - 1 - NT >= tile0) store_tile(s1 - 1 - NT);
}

// ---------------------------------------------------------------------------------------------------------------
// The Nyquist-slot quirk (Source.cpp:420-425; derivation in mx_kernels.hpp) for the fused kernel.  With wx, wy the alternating
// weights of a reflect-101 padded line (mx_alt_weight):
//     V'[r][x]  = V[r][x] + dr (-1)^(x+pad) Srow(r),            Srow(r) = sum_x wx(x) img[r][x]                       (integers)
//     out[r][x] = colconv(V')[r][x] + dc (-1)^(r+pad) Scol(x),  Scol(x) = sum_r wy(r) V'[r][x]
//                                                                       = rowconv(Ccol)(x) + dr (-1)^(x+pad) Z
//     Ccol(x) = sum_r wy(r) img[r][x]  (integers; the row convolution and the weighted column sum commute),  Z = sum_r wy(r) Srow(r)
// fx_altsums reads the image once and leaves Srow and, per band of rows, the partial Ccol; fx_quirk_terms turns them into
// qrow[f][c][re] = dr (-1)^pad Srow[refl(re - PADA)][c] and qcol[f][3 x + c] = dc (-1)^pad Scol, which the fused kernel adds
// as qrow (-1)^x (to V, before the hand-off) and qcol (-1)^r (to the output, before the truncation).
#ifdef BLUR_FX_QUIRK_KERNELS   // engine.hip only: plain (non-template) kernels must live in one translation unit
constexpr int kFxSumRows = 32;          // image rows per band (packed 16-bit column sums: 32 x 3 x 255 < 65536)

// sum over the 16 lanes of a DPP row, valid in every lane of the row
__device__ __forceinline__ int fx_row16_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);    // row_mirror
    return v;
}

// grid (bands of kFxSumRows rows, batches of 256 twelve-byte groups = 1024 pixel columns, frames), 256 threads: a thread owns one
// group (4 pixels) of every row of the band, eight rows of loads in flight; cols % 4 == 0, frames 4-byte aligned.
//   srow_part[f][batch][r][c]  sum over the batch's pixels of wx(x) img[r][x][c]
//   cpart[f][band][3 x + c]    sum over the band's rows of wy(r) img[r][x][c]
__global__ __launch_bounds__(256) void fx_altsums(const uint8_t* __restrict__ src, int* __restrict__ srow_part, int* __restrict__ cpart, int rows, int cols, int pad,
                                                  int nbands, int nbatches)
{
    __shared__ int sred[kFxSumRows][3];
    const int f = blockIdx.z, batch = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
    const uint8_t* img = src + static_cast<size_t>(f) * rows * cols * 3;
    const int groups = cols / 4, r0 = band * kFxSumRows, r1 = min(r0 + kFxSumRows, rows);
    const int gi = batch * 256 + tid, x = 4 * gi;
    const bool act = gi < groups;
    if (tid < kFxSumRows * 3) (&sred[0][0])[tid] = 0;
    __syncthreads();
    const int flip = (pad & 1) ? -1 : 1;
    const bool plain = x > pad && x + 3 < cols - 1 - pad;                 // no pixel of the group is mirrored: weights +-1 by parity
    int wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = act ? mx_alt_weight(x + q, cols, pad) : 0;
    uint32_t accp[6], accn[6];                                            // packed 16-bit sums of the rows with positive / negative wy
#pragma unroll
    for (int j = 0; j < 6; ++j) accp[j] = accn[j] = 0;
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const uint8_t* col0 = img + 12 * static_cast<size_t>(act ? gi : 0);
    for (int rb = r0; rb < r1; rb += 8) {
        u3 d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = min(rb + i, r1 - 1);
            d[i] = *reinterpret_cast<const u3*>(col0 + static_cast<size_t>(r) * cols * 3);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = rb + i;
            if (r < r1) {                                                    // uniform
                const int wy = mx_alt_weight(r, rows, pad);
                int s[3] = { 0, 0, 0 };
                if (plain) {
                    const int e0 = static_cast<int>(d[i][0] ^ 0x80808080u), e1 = static_cast<int>(d[i][1] ^ 0x80808080u), e2 = static_cast<int>(d[i][2] ^ 0x80808080u);
                    // bytes (pixel q, channel c) = 3 q + c of the group, signs + - + - over q; the offset 128 cancels (weights sum to 0)
                    int t0 = __builtin_amdgcn_sdot4(e0, static_cast<int>(0xff000001u), 0, false);
                    t0 = __builtin_amdgcn_sdot4(e1, 0x00010000, t0, false);
                    t0 = __builtin_amdgcn_sdot4(e2, 0x0000ff00, t0, false);
                    int t1 = __builtin_amdgcn_sdot4(e0, 0x00000100, 0, false);
                    t1 = __builtin_amdgcn_sdot4(e1, 0x010000ff, t1, false);
                    t1 = __builtin_amdgcn_sdot4(e2, 0x00ff0000, t1, false);
                    int t2 = __builtin_amdgcn_sdot4(e0, 0x00010000, 0, false);
                    t2 = __builtin_amdgcn_sdot4(e1, 0x0000ff00, t2, false);
                    t2 = __builtin_amdgcn_sdot4(e2, static_cast<int>(0xff000001u), t2, false);
                    s[0] = flip * t0; s[1] = flip * t1; s[2] = flip * t2;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int byte = 3 * q + c;
                            s[c] += wq[q] * static_cast<int>((d[i][byte >> 2] >> (8 * (byte & 3))) & 0xffu);
                        }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int t = fx_row16_sum(act ? s[c] : 0);
                    if ((tid & 15) == 0) atomicAdd(&sred[r - r0][c], t);
                }
                // column sums: bytes 0, 2 of each dword in one packed pair, bytes 1, 3 in the other; |wy| = 1, 2 or 3 (uniform)
                const uint32_t aw = static_cast<uint32_t>(wy < 0 ? -wy : wy);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint32_t lo = (d[i][j] & 0x00ff00ffu) * aw, hi = ((d[i][j] >> 8) & 0x00ff00ffu) * aw;
                    if (wy > 0) { accp[2 * j] += lo; accp[2 * j + 1] += hi; }
                    else { accn[2 * j] += lo; accn[2 * j + 1] += hi; }
                }
            }
        }
    }
    if (act) {
        int o[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            o[4 * j] = static_cast<int>(accp[2 * j] & 0xffffu) - static_cast<int>(accn[2 * j] & 0xffffu);
            o[4 * j + 2] = static_cast<int>(accp[2 * j] >> 16) - static_cast<int>(accn[2 * j] >> 16);
            o[4 * j + 1] = static_cast<int>(accp[2 * j + 1] & 0xffffu) - static_cast<int>(accn[2 * j + 1] & 0xffffu);
            o[4 * j + 3] = static_cast<int>(accp[2 * j + 1] >> 16) - static_cast<int>(accn[2 * j + 1] >> 16);
        }
        int4* dstp = reinterpret_cast<int4*>(cpart + (static_cast<size_t>(f) * nbands + band) * (3 * cols) + 12 * gi);
        dstp[0] = make_int4(o[0], o[1], o[2], o[3]);
        dstp[1] = make_int4(o[4], o[5], o[6], o[7]);
        dstp[2] = make_int4(o[8], o[9], o[10], o[11]);
    }
    __syncthreads();
    if (tid < (r1 - r0) * 3) srow_part[((static_cast<size_t>(f) * nbatches + batch) * rows + r0) * 3 + tid] = (&sred[0][0])[tid];
}

// grid (row blocks + column blocks, frames), 256 threads.  Row blocks: qrow for 256 rows of V each.  Column blocks: qcol for 256
// values e = 3 x + c each -- Ccol over the bands into LDS (with the taps' reach on both sides), the row convolution in double.
// Every column block sums Z itself (3 rows values, integers).   taps: 2 pad + 1 floats, centre at pad.
__global__ __launch_bounds__(256) void fx_quirk_terms(const int* __restrict__ srow_part, const int* __restrict__ cpart, const float* __restrict__ taps,
                                                      float* __restrict__ qrow, float* __restrict__ qcol, int rows, int cols, int pad, int pada, int qrows, int qpitch,
                                                      int nbands, int nbatches, int nrowblocks, float dr, float dc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fxq_lds[];
    __shared__ double zs[3][256];
    const int f = blockIdx.y, tid = threadIdx.x;
    const double sp = (pad & 1) ? -1.0 : 1.0;
    const int* sr = srow_part + static_cast<size_t>(f) * nbatches * rows * 3;
    auto srow = [&](int r, int c) {                  // Srow: the batches' parts added up (integers)
        int v = 0;
        for (int b = 0; b < nbatches; ++b) v += sr[(static_cast<size_t>(b) * rows + r) * 3 + c];
        return v;
    };
    if (static_cast<int>(blockIdx.x) < nrowblocks) {
        const int re = blockIdx.x * 256 + tid;
        if (re < qrows) {
            const int r = mx_refl(re - pada, rows);
            for (int c = 0; c < 3; ++c) qrow[(static_cast<size_t>(f) * 3 + c) * qrows + re] = static_cast<float>(static_cast<double>(dr) * sp * srow(r, c));
        }
        return;
    }
    double z[3] = { 0, 0, 0 };
    for (int r = tid; r < rows; r += 256) {
        const double wy = static_cast<double>(mx_alt_weight(r, rows, pad));
        for (int c = 0; c < 3; ++c) z[c] += wy * srow(r, c);
    }
    for (int c = 0; c < 3; ++c) zs[c][tid] = z[c];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) for (int c = 0; c < 3; ++c) zs[c][tid] += zs[c][tid + o];
        __syncthreads();
    }
    const int e0 = (blockIdx.x - nrowblocks) * 256;
    // pixels whose column sums this block's outputs read: [xa, xb] inside the image (reflect-101 maps into it)
    const int xa = max(0, e0 / 3 - pad), xb = min(cols - 1, (e0 + 255) / 3 + pad), nval = 3 * (xb - xa + 1);
    int* cc = reinterpret_cast<int*>(fxq_lds);
    const int* cp = cpart + static_cast<size_t>(f) * nbands * (3 * cols) + 3 * xa;
    for (int i = tid; i < nval; i += 256) {
        int sum = 0;
        for (int b = 0; b < nbands; ++b) sum += cp[static_cast<size_t>(b) * (3 * cols) + i];
        cc[i] = sum;
    }
    __syncthreads();
    const int e = e0 + tid;
    if (e >= qpitch) return;
    const int x = e / 3, c = e - 3 * x;
    float out = 0.f;
    if (x < cols) {
        double acc = 0;
        for (int t = -pad; t <= pad; ++t) acc += static_cast<double>(taps[t + pad]) * cc[3 * (mx_refl(x + t, cols) - xa) + c];
        const double sx = ((x + pad) & 1) ? -1.0 : 1.0;
        out = static_cast<float>(static_cast<double>(dc) * sp * (acc + static_cast<double>(dr) * sx * zs[c][0]));
    }
    qcol[static_cast<size_t>(f) * qpitch + e] = out;
}
#endif  // BLUR_FX_QUIRK_KERNELS

// ---- launcher ----------------------------------------------------------------------------------------------------------
struct FxEntry {
    int nkb;
    hipError_t (*blur_u8)(hipStream_t, const uint8_t* src, uint8_t* dst, const void* frags, FxGeom g, int num_cus, const float* qrow, const float* qcol, int qpitch);
};

template <int NKB> hipError_t fx_launch_u8(hipStream_t st, const uint8_t* src, uint8_t* dst, const void* frags, FxGeom g, int num_cus, const float* qrow,
                                           const float* qcol, int qpitch)
{
    using C = FxCfg<NKB>;
    const int chunks = (g.cols + kFxChunk - 1) / kFxChunk;
    const long long strips = static_cast<long long>(chunks) * g.nframes;
    if (strips <= 0) return hipSuccess;
    // segments per strip: a segment repeats NT steps of run-in, so as few as fill the chip about once
    int tps = ((g.ntiles + C::NT - 1) / C::NT) * C::NT;
    while (tps > C::NT && strips * ((g.ntiles + tps - 1) / tps) < static_cast<long long>(num_cus)) tps -= C::NT;
    // evenly sized segments
    int nseg = (g.ntiles + tps - 1) / tps;
    tps = (((g.ntiles + nseg - 1) / nseg + C::NT - 1) / C::NT) * C::NT;
    nseg = (g.ntiles + tps - 1) / tps;
    const long long ntasks = strips * nseg;
    const int per_xcd = static_cast<int>((ntasks + 7) / 8);
    const dim3 grid(static_cast<unsigned>(8 * per_xcd));
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fx_blur_u8<NKB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fx_blur_u8<NKB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (qrow)
        hipLaunchKernelGGL((fx_blur_u8<NKB, true>), grid, dim3(256), C::LDS, st, src, dst, static_cast<const mx_half8*>(frags), g, chunks, tps, nseg,
                           static_cast<int>(ntasks), qrow, qcol, qpitch);
    else
        hipLaunchKernelGGL((fx_blur_u8<NKB, false>), grid, dim3(256), C::LDS, st, src, dst, static_cast<const mx_half8*>(frags), g, chunks, tps, nseg,
                           static_cast<int>(ntasks), qrow, qcol, qpitch);
    return hipGetLastError();
}

#define BLUR_FX(NKB_)                                                                                       \
    namespace blur_amd {                                                                                    \
    const FxEntry* fx_entry_##NKB_()                                                                        \
    {                                                                                                       \
        static const FxEntry e = { NKB_, fx_launch_u8<NKB_> };                                              \
        return &e;                                                                                          \
    }                                                                                                       \
    }

}  // namespace blur_amd
