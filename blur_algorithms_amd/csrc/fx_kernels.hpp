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
    static constexpr int FOFF = QOFF + 2 * 96 * 4;                    // the column pass's Toeplitz fragments: [hi, lo][NKB][64] x 16 bytes
    static constexpr int LDS = FOFF + 2 * NKB * 1024;
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
                                                     int qpitch, const uint16_t* __restrict__ tilemap)
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
    // Reflect-101 along the rows is folded into the ROW pass's fragments: a tile whose window reaches over the image's left or
    // right edge has its own fragment set in which a mirrored tap is added to the tap of the pixel it mirrors (host_math.hpp:
    // fx_fragment_sets); window positions outside the image carry zero taps, whatever the loads return there.
    // tilemap[tile column] = set.  The row pass's fragments (hi and lo halves) stay in registers; the column pass's -- always
    // set 0: its border rows are real rows of V -- are read from LDS, one ds_read_b128 per use.
    mx_half8 th[NKB], tl[NKB];
    {
        const int tt = 4 * xc + wave, set = tt < (g.cols + 31) / 32 ? tilemap[tt] : 0;
        const mx_half8* fs = frags + static_cast<size_t>(set) * 2 * NKB * 64;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            th[kb] = fs[kb * 64 + lane];
            tl[kb] = fs[(NKB + kb) * 64 + lane];
        }
        for (int i = tid; i < 2 * NKB * 64; i += 256) reinterpret_cast<mx_half8*>(fx_lds + C::FOFF)[i] = frags[i];
    }
    const mx_half8* cfp = reinterpret_cast<const mx_half8*>(fx_lds + C::FOFF) + lane;
#define ch_(d) cfp[(d) * 64]
#define cl_(d) cfp[(NKB + (d)) * 64]

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
    // (the frame as a buffer resource: a group that starts left of the image or runs past the frame's end is out of bounds and
    // reads as zero; inside, a group past a row's end reads the next row's first pixels -- zero taps either way)
    const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, static_cast<uint32_t>(g.rows) * g.cols * 3u, kMxRsrcWord3);
    auto issue_chunk = [&](int s, int j) __attribute__((always_inline)) {
        const int row = tid >> 3, g0 = tid & 7;
        const int r = mx_refl(32 * s - PADA + row, g.rows);
        const uint32_t off = (static_cast<uint32_t>(r) * g.cols + static_cast<uint32_t>(x0 - PADA + 4 * g0)) * 3u;
#pragma unroll
        for (int k = 3 * j; k < 3 * j + 3 && k < PER; ++k) {
            typedef uint32_t u3 __attribute__((ext_vector_type(3)));
            const bool in = (C::GPR % 8 == 0) || k < PER - 1 || g0 < C::GPR % 8;
            const u3 t = __builtin_amdgcn_raw_buffer_load_b96(rimg, in ? off + 96u * k : off, 0, 0);
            raw.d[k][0] = t[0]; raw.d[k][1] = t[1]; raw.d[k][2] = t[2];
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
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    uint2 wd;
                    {   // pixels 0, 1: bytes c, 3 + c
                        const int B0 = c, B1 = 3 + c, da = B1 >> 2, db = B0 >> 2;
                        const uint32_t sel = fx_sel1(B0, da, db) | (0x0cu << 8) | (fx_sel1(B1, da, db) << 16) | (0x0cu << 24);
                        wd.x = __builtin_amdgcn_perm(raw.d[k][da], raw.d[k][db], sel);
                    }
                    {   // pixels 2, 3: bytes 6 + c, 9 + c
                        const int B0 = 6 + c, B1 = 9 + c, da = B1 >> 2, db = B0 >> 2;
                        const uint32_t sel = fx_sel1(B0, da, db) | (0x0cu << 8) | (fx_sel1(B1, da, db) << 16) | (0x0cu << 24);
                        wd.y = __builtin_amdgcn_perm(raw.d[k][da], raw.d[k][db], sel);
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
            const mx_half8 fh = ch_(d), fl = cl_(d);
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
            const mx_half8 fh = ch_(d), fl = cl_(d);
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
                __builtin_amdgcn_sched_barrier(0);
                if (c == 2) rowpass(cur ^ 1, 0); else rowpass(cur, c + 1);
                if (c == 2) { commit_chunk(cur, 0); issue_chunk(s + 3, 0); }
                else if (c == 0) { commit_chunk(cur ^ 1, 1); issue_chunk(s + 2, 1); }
                else { commit_chunk(cur ^ 1, 2); issue_chunk(s + 2, 2); }
                if (c == 0 && s - 1 - NT >= tile0 && s > s0) store_tile(s - 1 - NT);
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase B: this product's column pass; the next product's hand-off; this tile's bytes
                colpass(hb, c, qs);
                if (c == 2) split(hb ^ 1, cur ^ 1, 0); else split(hb ^ 1, cur, c + 1);
                __builtin_amdgcn_sched_barrier(0);
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

#undef ch_
#undef cl_
// ---- launcher ----------------------------------------------------------------------------------------------------------
struct FxEntry {
    int nkb;
    hipError_t (*blur_u8)(hipStream_t, const uint8_t* src, uint8_t* dst, const void* frags, const uint16_t* tilemap, FxGeom g, int num_cus, const float* qrow, const float* qcol,
                          int qpitch);
};

template <int NKB> hipError_t fx_launch_u8(hipStream_t st, const uint8_t* src, uint8_t* dst, const void* frags, const uint16_t* tilemap, FxGeom g, int num_cus,
                                           const float* qrow, const float* qcol, int qpitch)
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
                           static_cast<int>(ntasks), qrow, qcol, qpitch, tilemap);
    else
        hipLaunchKernelGGL((fx_blur_u8<NKB, false>), grid, dim3(256), C::LDS, st, src, dst, static_cast<const mx_half8*>(frags), g, chunks, tps, nseg,
                           static_cast<int>(ntasks), qrow, qcol, qpitch, tilemap);
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
