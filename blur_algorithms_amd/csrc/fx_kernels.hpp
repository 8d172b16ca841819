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
// rows and columns BEFORE a pixel can leave, so they come from a pre-pass over the image (fx_prepass: exact integers, complete
// when it ends), and the fused kernel turns them into its per-row and per-column terms itself (struct FxQuirk); the column sums
// of V the column term is made of are, by linearity, the row convolution of the image's weighted column sums.
#pragma once
#include "mx_kernels.hpp"
#include <atomic>

#if !defined(FX_NO_SGB) && !defined(FX_SGB)
#define FX_SGB 1     // ask the scheduler for the interleaved order of each phase (sched_group_barrier)
#endif
#ifndef FX_SGB_A
#define FX_SGB_A 5
#endif
#ifndef FX_SGB_B
#define FX_SGB_B 6
#endif

namespace blur_amd {

struct FxGeom {
    int rows, cols, pad;
    int nframes;
    int aligned;      // 1: cols % 4 == 0 and frame pointers 4-byte aligned: 12-byte pixel groups are three aligned dwords (informational: the
                      // kernels take any width and alignment; unaligned 12-byte accesses are what the memory pipeline then sees)
    int ntiles;       // output tiles of 32 rows per frame
    int nright;       // chunks at the right edge that read their window from a strip (the chunk at the left edge always does)
    int nxcd;         // XCDs of the device (hipDeviceAttributeNumberOfXccs): workgroup b runs on XCD b % nxcd
};

constexpr int kFxChunk = 128;     // pixel columns per workgroup
constexpr int kFxMaxBatches = 4;  // column batches of the quirk's pre-pass = parts of a row's alternating sum (fx_prepass)
// chunks at the left edge whose window (from 128 xc - pada) starts left of the image: they read a strip
__host__ __device__ constexpr int fx_left_strips(int pada) { return pada > 0 ? (pada + kFxChunk - 1) / kFxChunk : 1; }
// Which of a strip chunk's staging loads come from its strip (fx_blur_u8, round 4): thread (row, g0) loads the 12-byte groups g0 + 8 k,
// k = 0 .. per - 1, of a window row, and a load instruction has ONE buffer resource -- so the strip holds the groups of whole k:
// k in [*lo, *hi) (every group with a mirrored pixel lies in there), and the other k read the image itself.  For the 4K frame's
// left chunk that is 24 of 68 groups (72 mirrored pixels + 24 copied ones) instead of the whole 272-pixel window.
__host__ __device__ inline void fx_strip_range(int xc, int cols, int pada, int per, int* lo, int* hi)
{
    const int x0 = kFxChunk * xc;
    const bool left = x0 - pada < 0, right = x0 + kFxChunk + pada > cols;
    if (left && right) { *lo = 0; *hi = per; }
    else if (left) { const int gl = (pada - x0 + 3) / 4; const int h = (gl + 7) / 8; *lo = 0; *hi = h < per ? h : per; }     // groups below gl hold a pixel < 0
    else if (right) { const int gb = (cols - (x0 - pada)) / 4; *lo = gb / 8; *hi = per; }                                   // groups from gb on hold a pixel >= cols
    else { *lo = 0; *hi = 0; }
}

template <int NKB> struct FxCfg {
    static constexpr int PADA = 8 * (NKB - 2), WIN = kFxChunk + 2 * PADA, GPR = WIN / 4, PER = (GPR + 7) / 8;
    // halfs per LDS row of the window: every thread commits PER groups of 4 positions, 32 apart, WITHOUT a lane mask (an exec-masked
    // commit is a branch in the middle of a slice of matrix instructions: the slice's vector work then runs behind them instead of
    // beside them), so a row holds 32 PER positions; pitch = 4 mod 8 dwords as in mx_row_pitch (conflict-free ds_read_b128)
    static constexpr int fx_pitch() { int dw = 16 * PER; while ((dw & 7) != 4) ++dw; return 2 * dw; }
    static constexpr int PW = fx_pitch();
    static constexpr int NT = (NKB - 1) / 2;                          // live accumulator tiles per channel = steps per unrolled round
    static constexpr int BUF = 3 * 32 * PW * 2;                       // bytes of one window buffer
    static constexpr int QOFF = 2 * BUF;                              // qrow stage: [2 buffers][x even, odd: +q, -q][256] floats (96 used:
    static constexpr int LDS = QOFF + 2 * 2 * 256 * 4;                // [3][32]; every thread stores, threads 96 .. 255 into the padding)
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

#ifdef FX_ABL_M16
// timing-only ablation: every 32x32x16 product replaced by two 16x16x32 products on quarter tiles (the same matrix-pipe cycles and
// operand reads; the results are garbage) -- does the clock the chip grants depend on the shape?
typedef float fx_f4 __attribute__((ext_vector_type(4)));
struct fx_tile { fx_f4 q[4]; };
#define FX_EL(t, i) ((t).q[(i) >> 2][(i) & 3])
#define FX_PIN(t) asm volatile("" : "+a"((t).q[0]), "+a"((t).q[1]), "+a"((t).q[2]), "+a"((t).q[3]))
__device__ __forceinline__ void fx_mma(mx_half8 a, mx_half8 b, fx_tile& t, int pair)
{
    t.q[2 * pair] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, t.q[2 * pair], 0, 0, 0);
    t.q[2 * pair + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, t.q[2 * pair + 1], 0, 0, 0);
}
#else
typedef mx_float16 fx_tile;
#define FX_EL(t, i) ((t)[i])
#define FX_PIN(t) asm volatile("" : "+a"(t))
__device__ __forceinline__ void fx_mma(mx_half8 a, mx_half8 b, fx_tile& t, int) { t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, t, 0, 0, 0); }
#endif

// 4 x 4 byte transpose inside a lane quad: lane j holds bytes M[j][0..3]; afterwards lane k holds M[0..3][k]
__device__ __forceinline__ uint32_t fx_quad_transpose(uint32_t p, uint32_t sel1, uint32_t sel2)
{
    const uint32_t t = fx_dpp<0xb1>(p);                       // quad_perm [1,0,3,2]
    const uint32_t q = __builtin_amdgcn_perm(t, p, sel1);
    const uint32_t u = fx_dpp<0x4e>(q);                       // quad_perm [2,3,0,1]
    return __builtin_amdgcn_perm(u, q, sel2);
}

// What the fused kernels need for the Nyquist-slot quirk (Source.cpp:420-425; the algebra is at fx_altsums_body): the pre-pass's
// exact integer partial sums and the taps.  The kernels turn them into their terms themselves:
//   row term      qrow(r, c) = dr (-1)^pad Srow(r, c), added to V as qrow (-1)^x           (the row's batch parts: a few loads per row)
//   column term   qcol(x, c) = dc (-1)^pad (rowconv(Ccol)(x, c) + dr (-1)^(x+pad) Z(c)), added to the output as qcol (-1)^r: a
//                 workgroup adds up the bands' parts of the 128 + 2 pad values of Ccol around its own chunk and convolves them in
//                 its prologue (fx_quirk_cols_tile)
// so a call is two launches (pre-pass, fused kernel); rounds 2-3 had two more kernels for the terms between them (13 us and two
// launch gaps per call; the prologue costs the fused kernel a few us).
struct FxQuirk {
    const int* srow_part;       // [frame][batch][row][3]    parts of Srow(r, c) = sum_x wx(x) img[r][x][c]
    const int* cpart;           // [frame][band][3 x + c]    parts of Ccol(x, c) = sum_r wy(r) img[r][x][c]
    const long long* zpart;     // [frame][band][batch][3]   parts of Z(c) = sum_r wy(r) Srow(r, c)
    const float* taps;          // the 2 pad + 1 taps of the row pass, centre at pad
    int nbatches, nbands;
    int cpitch;                 // ints per band row of cpart: 12 x groups of 4 pixels (>= 3 cols)
    float dr, dc;
};

// qc[NCH xl + c] (xl = 0 .. 127; NCH = 3: c = 0 .. 2, NCH = 1: channel c0 only) = the column term of pixel x0 + xl, 0 right of the
// image.  256 threads; `scratch` = LDS for NCH (128 + 2 pad) + 2 pad + 1 + 6 doubles and 3 (128 + 2 pad) + 4 ints; ends with a barrier,
// after which scratch is free again and qc is valid.  One wave per SIMD hides no latency, and a workgroup's 128 + 2 pad columns x 3
// channels x nbands parts are 26 000 values (4K, sigma 20): one dword per lane and load kept the CU's address unit busy for 3 us per
// trip of 48 loads (round 4, first form).  So: the parts are read 16 bytes per lane -- the values of the image columns the window
// maps to under reflect-101 are CONTIGUOUS in a band's row of cpart --, 17 bands of independent loads per trip, summed as integers
// into LDS; the window (reflect-101 applied, converted to double ONCE) is filled from there; the convolution runs in double, eight
// taps per trip.
template <int NCH>
__device__ __forceinline__ void fx_quirk_cols_tile(unsigned char* scratch, float* qc, const FxQuirk& q, int f, int x0, int c0, int cols, int pad, int tid)
{
    const int win = kFxChunk + 2 * pad, nval = NCH * win, ntap = 2 * pad + 1;
    double* cc = reinterpret_cast<double*>(scratch);
    double* tp = cc + nval;
    double* zs = tp + ntap;
    int* S = reinterpret_cast<int*>(zs + 4 + ((nval + ntap) & 1));          // 16-byte aligned
    const size_t bstride = static_cast<size_t>(q.cpitch);
    // the image columns [xlo, xhi) that the window positions a valid output reads (x0 - pad .. min(x0 + 127, cols - 1) + pad) map to
    const int wlo = x0 - pad, whi = min(x0 + kFxChunk, cols) + pad;
    int xlo = max(wlo, 0), xhi = min(whi, cols);
    if (wlo < 0) xhi = max(xhi, min(cols, 1 - wlo));                        // position x < 0 reads column -x
    if (whi > cols) xlo = min(xlo, max(0, 2 * cols - 1 - whi));             // position x >= cols reads column 2 (cols - 1) - x
    {
        typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
        typedef int i4 __attribute__((ext_vector_type(4)));
        constexpr int BG = 17;
        const int quads = (3 * (xhi - xlo) + 3) / 4;                        // (the last one may read up to 12 bytes of the next row / of zpart)
        const int* base = q.cpart + static_cast<size_t>(f) * q.nbands * bstride + 3 * xlo;
        for (int qd = tid; qd < quads; qd += 256) {
            const int* cp = base + 4 * qd;
            i4 sum = { 0, 0, 0, 0 };
            for (int b0 = 0; b0 < q.nbands; b0 += BG) {
                i4 t[BG];
#pragma unroll
                for (int j = 0; j < BG; ++j) t[j] = *reinterpret_cast<const i4u*>(cp + static_cast<size_t>(min(b0 + j, q.nbands - 1)) * bstride);
#pragma unroll
                for (int j = 0; j < BG; ++j)
                    if (b0 + j < q.nbands) sum += t[j];
            }
            *reinterpret_cast<i4*>(S + 4 * qd) = sum;
        }
    }
    for (int i = tid; i < ntap; i += 256) tp[i] = static_cast<double>(q.taps[i]);
    {   // Z: wave w < 3 adds up channel w's parts (exact integers), four independent loads per lane and trip, then across the wave
        const int w = tid >> 6, l = tid & 63, nz = q.nbands * q.nbatches;
        if (w < 3 && (NCH == 3 || w == c0)) {
            const long long* zp = q.zpart + static_cast<size_t>(f) * nz * 3 + w;
            long long z = 0;
            for (int i0 = l; i0 < nz; i0 += 256) {
                long long t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = zp[static_cast<size_t>(min(i0 + 64 * j, nz - 1)) * 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) z += i0 + 64 * j < nz ? t[j] : 0;
            }
            for (int o = 32; o >= 1; o >>= 1) z += __shfl_xor(z, o, 64);
            if (l == 0) zs[w] = static_cast<double>(z);
        }
    }
    __syncthreads();
    for (int i = tid; i < nval; i += 256) {
        const int p = i / NCH, ch = NCH == 3 ? i - 3 * p : c0;
        const int xr = min(max(mx_refl(x0 - pad + p, cols), xlo), xhi - 1);     // (clamped: positions no valid output reads)
        cc[i] = static_cast<double>(S[3 * (xr - xlo) + ch]);
    }
    __syncthreads();
    const double sp = (pad & 1) ? -1.0 : 1.0;
    for (int e = tid; e < NCH * kFxChunk; e += 256) {
        const int xl = e / NCH, c = NCH == 3 ? e - 3 * xl : c0, x = x0 + xl;
        float out = 0.f;
        if (x < cols) {
            const double* ccx = cc + NCH * xl + (NCH == 3 ? c : 0);           // tap t = -pad sits here
            double acc[4] = { 0, 0, 0, 0 };
            int t = 0;
            for (; t + 8 <= ntap; t += 8) {
                double a[8], v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j] = tp[t + j]; v[j] = ccx[NCH * (t + j)]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_fma(a[j], v[j], acc[j & 3]);
            }
            for (; t < ntap; ++t) acc[t & 3] = __builtin_fma(tp[t], ccx[NCH * t], acc[t & 3]);
            const double sx = ((x + pad) & 1) ? -1.0 : 1.0;
            out = static_cast<float>(static_cast<double>(q.dc) * sp * (((acc[0] + acc[1]) + (acc[2] + acc[3])) + static_cast<double>(q.dr) * sx * zs[c]));
        }
        qc[e] = out;
    }
    __syncthreads();
}

// One workgroup per (frame, segment of output tiles, chunk of 128 pixel columns).
//   qk (QUIRK)       the pre-pass's integer sums: the row pass's term qrow (-1)^x per row of V and the column pass's term qcol (-1)^r
//                    per output column are made from them here (struct FxQuirk)
//
// Software pipeline.  n = (step s, channel c) enumerates the wave's (32 rows x 32 pixels) products; per n the matrix pipe runs
//     phase A(n):  R(n+1)  row pass of the NEXT product (2 NKB MFMAs)            beside: staging of later windows, the stores
//     phase B(n):  C(n)    column pass of this one (3 NKB MFMAs)                 beside: hand-off S(n+1), emission E(n)
// so every vector instruction has matrix work of another product to hide behind, and exactly 16 accumulator tiles are live
// (15 of the column pass + the row pass's): C(n) begins with the tile that FINISHES (its last window block) and ends with the
// tile that STARTS, which takes over the finished tile's registers after E(n) has read them; the row accumulator is read out
// by S(n+1) early in B(n), before R(n+2) writes it again.
// DUMPV (tests only, blur_rowpass_u8c3_dev with BLUR_ENGINE_FUSED): the row pass's float planes V' (quirk term included), as the
// hand-off reads them, also go to vdump[frame][channel][row][col] -- what the reference holds in `resf` after Source.cpp:520-537.
// RAGGED: the image width is not a multiple of 4: the last quad of 4 pixels of a row holds 1 .. 3 of them and leaves as single bytes
// (nine more store instructions per row group, with an offset outside the buffer for every lane that has nothing to store)
template <int NKB, bool QUIRK, bool DUMPV = false, bool RAGGED = false>
__global__ __launch_bounds__(256, 1) void fx_blur_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const mx_half8* __restrict__ frags, FxGeom g,
                                                     int chunks, int tps, int nseg, int ntasks, FxQuirk qk, const uint8_t* __restrict__ strips,
                                                     float* __restrict__ vdump)
{
    using C = FxCfg<NKB>;
    constexpr int PADA = C::PADA, PW = C::PW, NT = C::NT, PER = C::PER;
    static_assert(PER <= 9, "three staging chunks of three 12-byte groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char fx_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;

    // tasks are numbered chunk-fastest; workgroups b and b + 8 share an XCD: every XCD gets a contiguous band of tasks, so the
    // strips left and right of a strip -- which read 2 PADA of its window columns -- run on the same L2 at about the same time
    const int nx = g.nxcd, xcd = blockIdx.x % nx, in_xcd = blockIdx.x / nx, per_xcd = (ntasks + nx - 1) / nx, task = xcd * per_xcd + in_xcd;
    if (in_xcd >= per_xcd || task >= ntasks) return;
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
    const bool in_cols = xpix + 3 < g.cols;                        // the whole quad lies inside the image: one 12-byte store
    const int tail_bytes = RAGGED && !in_cols && xpix < g.cols ? 3 * (g.cols - xpix) : 0;       // 3, 6 or 9 bytes of a quad cut by the right edge
    float cpos[3], cneg[3];          // set in the prologue below, after the first window's loads are on their way
    const int qrows = 32 * (g.ntiles + NT);
    const double qrs = QUIRK ? static_cast<double>(qk.dr) * ((g.pad & 1) ? -1.0 : 1.0) : 0.0;

#ifdef FX_ABL_M16
    const fx_f4 zero4 = { 0.f, 0.f, 0.f, 0.f };
    const fx_tile zero = { { zero4, zero4, zero4, zero4 } };
#else
    const fx_tile zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#endif
    fx_tile acc[3][NT];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[c][k] = zero;
    fx_tile arow = zero;             // the row pass's accumulator
    uint32_t hl[2][8];                  // hand-off: [hi, lo][packed row pairs], block b = entries 4 b .. 4 b + 3
    uint32_t rr[3][4];                  // finished tile, per channel and row group: 4 pixels of one row (after the quad transpose)

    // step s handles rows 32 s .. 32 s + 31 of V (row re of V = image row refl(re - PADA)) and emits tile s - NT
    const int s0 = tile0, s1 = tile1 + NT;
    FxRaw<NKB> raw;
    int qpart[kFxMaxBatches] = {};
    // staging in three chunks of three 12-byte groups: commit chunk j of window s, then refill the registers with window s + 1
    // Reflect-101 along the rows (Source.cpp:525-529) never shows in this kernel: a chunk whose window reaches over the image's left
    // or right edge reads it from a STRIP -- a copy of its 128 + 2 PADA window columns with the mirrored pixels in place, written by
    // fx_edge_strips before the launch (left + nright strips per frame) -- and every other chunk's window lies inside the image.  So a
    // window row is 17 x 4 twelve-byte groups at base + row * pitch for every workgroup; the source is a buffer resource (groups
    // past the end of the last row read as zero; they only meet zero taps).
    // Round 4: the strip of such a chunk holds only the staging loads (whole k) that touch a mirrored pixel -- fx_strip_range: k in
    // [ka_lo, ka_hi) -- and the chunk's other loads read the image like everybody else's (two resources, chosen per load
    // instruction by a wave-uniform test): a third of the strips' bytes, which the pre-pass writes at the memory's copy rate.
    constexpr int NLEFT = fx_left_strips(PADA);
    const int sidx = xc < NLEFT ? xc : (xc >= chunks - g.nright ? NLEFT + xc - (chunks - g.nright) : -1);      // uniform
    int ka_lo = 0, ka_hi = 0;
#ifdef FX_FULL_STRIPS
    if (sidx >= 0) ka_hi = PER;
#else
    if (sidx >= 0) fx_strip_range(xc, g.cols, PADA, PER, &ka_lo, &ka_hi);
#endif
    const uint32_t pitch_s = 3u * C::WIN, pitch_i = 3u * static_cast<uint32_t>(g.cols);
    const uint8_t* ibase = img + 3 * (x0 - PADA);               // (before the frame for the left chunk: its loads below ka_hi never use it)
    const uint32_t ibytes = (static_cast<uint32_t>(g.rows) * g.cols - static_cast<uint32_t>(x0 - PADA)) * 3u;
    const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ibase), 0, ibytes, kMxRsrcWord3);
    const __amdgpu_buffer_rsrc_t rstrip = sidx >= 0 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(strips + (static_cast<size_t>(f) * (NLEFT + g.nright) + sidx) * g.rows * (3 * C::WIN)),
                                                                                        0, static_cast<uint32_t>(g.rows) * 3u * C::WIN, kMxRsrcWord3) : rimg;
    // staging map: a thread owns the 12-byte groups g0 + 8 k of window row `srow`.  ds_write_b64 is served in groups of 16 lanes over 32
    // banks: the two rows of a group must lie 16 banks apart, and the row pitch is 4 mod 8 dwords, so they are rows a and a + 4
    // (a wave still covers 8 consecutive rows: its loads are 8 rows x 96 contiguous bytes as before)
#ifdef FX_STAGE_LINEAR
    const int srow = tid >> 3;
#else
    const int srow = 8 * (tid >> 6) + ((tid >> 4) & 3) + 4 * ((tid >> 3) & 1);
#endif
    auto issue_chunk = [&](int s, int j) __attribute__((always_inline)) {
        const int row = srow, g0 = tid & 7;
#ifdef FX_ABL_NOLOAD
        if (s >= 0) return;
#endif
        const int r = mx_refl(32 * s - PADA + row, g.rows);
        const uint32_t off_i = static_cast<uint32_t>(r) * pitch_i + 12u * g0, off_s = static_cast<uint32_t>(r) * pitch_s + 12u * g0;
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
#pragma unroll
        for (int k = 3 * j; k < 3 * j + 3 && k < PER; ++k) {
            const bool in = (C::GPR % 8 == 0) || k < PER - 1 || g0 < C::GPR % 8;
            const bool from_strip = k >= ka_lo && k < ka_hi;                                 // wave-uniform
            const uint32_t off = from_strip ? off_s : off_i;
            // (lanes whose group lies past the window's GPR -- the last k only -- repeat the group of the k before: a valid address in
            // either resource; the row's first groups are not, for the left chunk's image resource: it starts before the frame)
            const u3 t = __builtin_amdgcn_raw_buffer_load_b96(from_strip ? rstrip : rimg, off + 96u * (in ? k : (k > 0 ? k - 1 : 0)), 0, 0);
            raw.d[k][0] = t[0]; raw.d[k][1] = t[1]; raw.d[k][2] = t[2];
        }
        if (QUIRK && j == 0) {
            // the row term of row re of V (= image row refl(re - PADA)): dr (-1)^pad Srow, rounded once as the term kernels of rounds 2-3 did
            // (threads 0 .. 95 = (channel, row); the others repeat channel 2: no lane mask, see FxCfg)
            const int c = min(tid >> 5, 2), re = min(32 * s + (tid & 31), qrows - 1);
            const int* sp = qk.srow_part + (static_cast<size_t>(f) * qk.nbatches * g.rows + mx_refl(re - PADA, g.rows)) * 3 + c;
            // the row's batch parts (at most kFxMaxBatches: fx_prepass widens its batches for wide images): independent loads, no
            // loop with a wait inside -- they are added up when the term is committed, a step later
#pragma unroll
            for (int b = 0; b < kFxMaxBatches; ++b) qpart[b] = sp[static_cast<size_t>(min(b, qk.nbatches - 1)) * g.rows * 3];
        }
    };
    // one (group k, channel c) of the window: two v_perm_b32 (deinterleave + the low byte of a binary16 each) and one ds_write_b64
    auto commit_item = [&](int buf, int k, int c) __attribute__((always_inline)) {
#ifdef FX_ABL_NOCOMMIT
        if (buf >= 0) return;
#endif
        const int row = srow, g0 = tid & 7;
        if (k >= PER) return;
        {   // (groups past the window's GPR of the last k hold whatever their clamped load returned: they land in the row's padding)
            _Float16* base = reinterpret_cast<_Float16*>(fx_lds + buf * C::BUF) + row * PW + 4 * g0;
            // window pixels (0, 1) and (2, 3) of the group: bytes (c, 3 + c) and (6 + c, 9 + c)
            auto pick = [&](int B0, int B1) __attribute__((always_inline)) {
                const int da = B1 >> 2, db = B0 >> 2;
                const uint32_t sel = fx_sel1(B0, da, db) | (0x0cu << 8) | (fx_sel1(B1, da, db) << 16) | (0x0cu << 24);
                return __builtin_amdgcn_perm(raw.d[k][da], raw.d[k][db], sel);
            };
            uint2 wd;
#ifdef FX_ABL_NOPERM
            wd.x = raw.d[k][c]; wd.y = raw.d[k][(c + 1) % 3];
#else
            wd.x = pick(c, 3 + c);
            wd.y = pick(6 + c, 9 + c);
#endif
#ifdef FX_ABL_NODSWRITE
            asm volatile("" ::"v"(wd.x), "v"(wd.y));
#else
            *reinterpret_cast<uint2*>(base + c * 32 * PW + 32 * k) = wd;
#endif
        }
    };
    auto commit_q = [&](int buf) __attribute__((always_inline)) {
        if (QUIRK) {
            int v = qpart[0];
#pragma unroll
            for (int b = 1; b < kFxMaxBatches; ++b) v += b < qk.nbatches ? qpart[b] : 0;
            const float qraw = static_cast<float>(qrs * v);
            float* qs = reinterpret_cast<float*>(fx_lds + C::QOFF) + buf * 512 + tid;
            qs[0] = qraw;            // the term enters as qrow (-1)^x: lanes of even x read this copy,
            qs[256] = -qraw;         // lanes of odd x this one
        }
    };
    auto commit_chunk = [&](int buf, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 9; ++i) commit_item(buf, 3 * j + i / 3, i % 3);
        if (j == 0) commit_q(buf);
    };
    // R: 32 rows x 32 pixels of channel c of the window in buffer `buf` -> arow
    // `beside(kb)` runs between the products of window block kb and kb + 1: the phase's vector work is handed out in slices, and a
    // scheduling fence after every block keeps the slices where they are (a window read is in flight for three blocks)
    auto rowpass = [&](int buf, int c, auto beside) __attribute__((always_inline)) {
        const _Float16* base = reinterpret_cast<const _Float16*>(fx_lds + buf * C::BUF) + (c * 32 + m) * PW + wave * 32 + 8 * h;
        fx_tile a = zero;
        mx_half8 x[4];
#pragma unroll
#ifdef FX_ABL_NOXREAD
        for (int kb = 0; kb < 4; ++kb) x[kb] = th[kb];
#else
        for (int kb = 0; kb < 3 && kb < NKB; ++kb) x[kb] = *reinterpret_cast<const mx_half8*>(base + 16 * kb);
#endif
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#ifndef FX_ABL_NOXREAD
            if (kb + 3 < NKB) x[(kb + 3) & 3] = *reinterpret_cast<const mx_half8*>(base + 16 * (kb + 3));
#endif
            fx_mma(x[kb & 3], th[kb], a, 0);
            fx_mma(x[kb & 3], tl[kb], a, 1);
            FX_PIN(a);                           // pins the two products between this block's fences (they have no other side effect)
            beside(kb);
#ifdef FX_SGB
            // inside the block: the slice's vector work after each product, not behind both
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_A, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_A, 0);
#endif
#ifndef FX_NOSB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        arow = a;
    };
    // S: arow -> scale (+ quirk), split into hi + lo, exchange with lane ^ 32 -> hl; in eight pieces (two halves of the 32 rows x
    // {read + scale -> sv, convert two row pairs, convert the other two, exchange}).  Phase B(n) runs the two read pieces of product n + 1 (the row accumulator is
    // free again before R(n + 2) starts), phase A(n + 1) the other four: B is the phase whose vector work fills its matrix time
    float sv[16];
    auto split_piece = [&](int buf, int c, int piece, int sprod = 0) __attribute__((always_inline)) {
        const int hf = piece / 4, sub = piece % 4;       // sub: 0 read + scale, 1 / 2 convert row pairs 0, 1 / 2, 3, 3 exchange
        uint32_t (&hp)[8] = hl[0];
        uint32_t (&lp)[8] = hl[1];
#ifdef FX_ABL_NOSPLIT
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { hp[4 * hf + k] = __builtin_bit_cast(uint32_t, FX_EL(arow, 8 * hf + k)); lp[4 * hf + k] = __builtin_bit_cast(uint32_t, FX_EL(arow, 8 * hf + 4 + k)); }
        }
        return;
#endif
        if (sub == 0) {
            if (QUIRK) {
                const float* qs4 = reinterpret_cast<const float*>(fx_lds + C::QOFF) + buf * 512 + (m & 1) * 256 + c * 32 + 4 * h;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float4 t4 = *reinterpret_cast<const float4*>(qs4 + 8 * (2 * hf + k));
                    sv[8 * hf + 4 * k] = __builtin_fmaf(FX_EL(arow, 8 * hf + 4 * k), kFxRowUnscale, t4.x);
                    sv[8 * hf + 4 * k + 1] = __builtin_fmaf(FX_EL(arow, 8 * hf + 4 * k + 1), kFxRowUnscale, t4.y);
                    sv[8 * hf + 4 * k + 2] = __builtin_fmaf(FX_EL(arow, 8 * hf + 4 * k + 2), kFxRowUnscale, t4.z);
                    sv[8 * hf + 4 * k + 3] = __builtin_fmaf(FX_EL(arow, 8 * hf + 4 * k + 3), kFxRowUnscale, t4.w);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) sv[8 * hf + k] = FX_EL(arow, 8 * hf + k) * kFxRowUnscale;
            }
            if (DUMPV) {
                const int x = x0 + 32 * wave + m;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int reg = 8 * hf + k, r = 32 * sprod + (reg & 3) + 8 * (reg >> 2) + 4 * h - PADA;
                    if (r >= 0 && r < g.rows && x < g.cols) vdump[((static_cast<size_t>(f) * 3 + c) * g.rows + r) * g.cols + x] = sv[8 * hf + k];
                }
            }
        } else if (sub == 1 || sub == 2) {
#pragma unroll
            for (int k = 2 * (sub - 1); k < 2 * sub; ++k) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const f2 vv = { sv[8 * hf + 2 * k], sv[8 * hf + 2 * k + 1] };
                hp[4 * hf + k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(vv, h2));
                float r0, r1;
                mx_remainder(hp[4 * hf + k], vv[0], vv[1], r0, r1);
                const f2 rem = { r0, r1 };
                lp[4 * hf + k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, h2));
            }
        } else {
            // regs (0..3, 4..7) of the half = rows (0..3, 8..11) + 4 h of its 16-row block -> the lane wants rows 8 h .. 8 h + 7
#ifndef FX_ABL_NOSWAP
            fx_swap4(hp[4 * hf], hp[4 * hf + 2], hp[4 * hf + 1], hp[4 * hf + 3], lp[4 * hf], lp[4 * hf + 2], lp[4 * hf + 1], lp[4 * hf + 3]);
#endif
        }
    };
    // E: one row group (4 rows) of the finished tile -> bytes, 4 x 4 transposed inside the lane quads -> rr[c][gq]
    fx_tile tfin = zero;
    auto emit_piece = [&](int c, int gq) __attribute__((always_inline)) {
#ifdef FX_ABL_NOEMIT
        rr[c][gq] = __builtin_bit_cast(uint32_t, FX_EL(tfin, 5 * gq));
        return;
#endif
        float fv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int reg = 4 * gq + k;
            fv[k] = __builtin_fmaf(FX_EL(tfin, reg), kMxUnscale, (reg & 1) ? cneg[c] : cpos[c]);
        }
        // (uint8_t)(v + 0.5f) of the reference (Utils.hpp:189,204-206): truncate, keep the low byte (values past 255.5, which only
        // the quirk's terms reach, wrap as on x86) -- the conversion writes its byte of the packed dword itself (SDWA)
        uint32_t pk = static_cast<uint32_t>(static_cast<int>(fv[0]));
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(pk) : "v"(fv[1]));
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(pk) : "v"(fv[2]));
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(pk) : "v"(fv[3]));
        rr[c][gq] = fx_quad_transpose(pk, sel1, sel2);
    };
    // C (+ S of the next product, + E): column pass of (step slot qs, channel c) from hl.  NKB triples of products: first the
    // tile that FINISHES with block 0 of this step (its window block NKB - 1), then blocks 2 .. NKB - 2 of the tiles in flight,
    // last blocks 0 and 1 of the tile that STARTS -- into the finished tile's registers, which E has read by then.  `beside(it)`
    // runs after triple it; a scheduling fence after every triple keeps the slices where they are.
    // `ri` (run-in): step ri of a segment's first NT (-1: any later step) -- tile s - a2 exists only for a2 <= ri, and the triples of
    // the others are left out STATICALLY (the first NT steps are their own copy of the step body: a run-time test inside a slice
    // would end the scheduling region; measured, 4.13 -> 5.53 us per step)
    auto colpass = [&](int c, int qs, int ri, auto beside) __attribute__((always_inline)) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        mx_half8 v1[2], v2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const u4 w1 = { hl[0][4 * b], hl[0][4 * b + 1], hl[0][4 * b + 2], hl[0][4 * b + 3] };
            const u4 w2 = { hl[1][4 * b], hl[1][4 * b + 1], hl[1][4 * b + 2], hl[1][4 * b + 3] };
            v1[b] = __builtin_bit_cast(mx_half8, w1);
            v2[b] = __builtin_bit_cast(mx_half8, w2);
        }
#pragma unroll
        for (int it = 0; it < NKB; ++it) {
            const int d = it == 0 ? NKB - 1 : (it >= NKB - 2 ? it - (NKB - 2) : it + 1);
            const int b = d & 1, a2 = d >> 1, slot = (qs - a2 + 2 * NT) % NT;
            if (ri < 0 || a2 <= ri) {
                fx_tile t = d == 0 ? zero : acc[c][slot];
                fx_mma(th[d], v1[b], t, 0);
                fx_mma(tl[d], v1[b], t, 1);
                fx_mma(th[d], v2[b], t, 0);
                FX_PIN(t);                       // pins the three products between this triple's fences
                if (it == 0) tfin = t; else acc[c][slot] = t;
            }
            beside(it);
#ifdef FX_SGB
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_B, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_B, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, FX_SGB_B, 0);
#endif
#ifndef FX_NOSB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    // F: the finished tile (rr) -> 12 interleaved bytes per lane and row group -> memory.  Buffer stores: rows past the image fall
    // outside the resource and are dropped by its bounds check; lanes whose quad lies right of the image, and tiles that do not
    // exist (run-in of a segment), get an offset outside it -- no branch.
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, static_cast<uint32_t>(g.rows) * g.cols * 3u, kMxRsrcWord3);
    const uint32_t lane_out = (static_cast<uint32_t>(4 * h + q) * g.cols + static_cast<uint32_t>(xpix)) * 3u;
    auto store_group = [&](int tile, bool valid, int gq) __attribute__((always_inline)) {
        {
            const uint32_t r0 = rr[0][gq], r1 = rr[1][gq], r2 = rr[2][gq];
            // lane q of the quad holds row q: r_c = channel c of pixels 0..3 -> 12 interleaved bytes
            const uint32_t X = __builtin_amdgcn_perm(r1, r0, 0x05010400u);      // [r0.0, r1.0, r0.1, r1.1]
            const uint32_t Y = __builtin_amdgcn_perm(r1, r0, 0x07030602u);      // [r0.2, r1.2, r0.3, r1.3]
            const uint32_t w0 = __builtin_amdgcn_perm(r2, X, 0x02040100u);      // [X0, X1, r2.0, X2]
            const uint32_t Z = __builtin_amdgcn_perm(r2, X, 0x0c0c0503u);       // [X3, r2.1, 0, 0]
            const uint32_t w1 = __builtin_amdgcn_perm(Y, Z, 0x05040100u);       // [Z0, Z1, Y0, Y1]
            const uint32_t w2 = __builtin_amdgcn_perm(r2, Y, 0x07030206u);      // [r2.2, Y2, Y3, r2.3]
            typedef uint32_t u3 __attribute__((ext_vector_type(3)));
            const u3 w = { w0, w1, w2 };
            const uint32_t rowoff = static_cast<uint32_t>(32 * tile + 8 * gq) * g.cols * 3u;        // uniform
#ifdef FX_ABL_NOSTORE
            const bool ok = false;
#else
            const bool ok = valid && in_cols && 32 * tile + 8 * gq + 4 * h + q < g.rows;
#endif
            __builtin_amdgcn_raw_buffer_store_b96(w, rout, ok ? lane_out + rowoff : 0xfffffff0u, 0, 0);
            if (RAGGED) {
                const bool rok = valid && 32 * tile + 8 * gq + 4 * h + q < g.rows;
#pragma unroll
                for (int b = 0; b < 9; ++b) {
                    const uint32_t word = b < 4 ? w0 : (b < 8 ? w1 : w2);
                    __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(word >> (8 * (b & 3))), rout, (rok && b < tail_bytes) ? lane_out + rowoff + b : 0xfffffff0u, 0, 0);
                }
            }
        }
    };
    auto store_tile = [&](int tile, bool valid) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) store_group(tile, valid, gq);
    };

    // prologue: window s0 and chunk 0 of window s0 + 1 in LDS; in the registers chunks 1, 2 of window s0 + 1 and chunk 0 of
    // window s0 + 2, as the loop expects them; R and S of the first product done
#pragma unroll
    for (int j = 0; j < 3; ++j) issue_chunk(s0, j);
    // (the quirk's column term while those loads travel: it works in LDS that the commits below overwrite)
    if (QUIRK) {
        // the column term of this chunk's pixels (nothing else uses LDS yet: tile and taps in window buffer 0, the result in buffer 1)
        static_assert(C::BUF >= 8 * (3 * C::WIN + 2 * PADA + 1 + 6) + 4 * (3 * C::WIN + 4), "fx_quirk_cols_tile's scratch fits window buffer 0");
        float* qc = reinterpret_cast<float*>(fx_lds + C::BUF);
        fx_quirk_cols_tile<3>(fx_lds, qc, qk, f, x0, 0, g.cols, g.pad, tid);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = qc[3 * (32 * wave + m) + c];
            cpos[c] = 0.5f + v;
            cneg[c] = 0.5f - v;
        }
        __syncthreads();                                   // before the staging writes windows over it
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) cpos[c] = cneg[c] = 0.5f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) commit_chunk(0, j);
    issue_chunk(s0 + 1, 0);
    commit_chunk(1, 0);
    issue_chunk(s0 + 1, 1);
    issue_chunk(s0 + 1, 2);
    issue_chunk(s0 + 2, 0);
    __syncthreads();
    rowpass(0, 0, [](int) {});
    split_piece(0, 0, 0, s0);
    split_piece(0, 0, 4, s0);

#ifdef FX_STAMPS
    // timing-only build: cycles (s_memtime) per phase kind, summed over the steps, by wave 0 of workgroup 0 -> the buffer passed as vdump
    unsigned long long st_acc[6] = { 0, 0, 0, 0, 0, 0 }, st_prev = __builtin_amdgcn_s_memtime(), st_begin = st_prev;
#define FX_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; }
#else
#define FX_STAMP(i)
#endif
    auto step = [&](int s, int qs, int ri) __attribute__((always_inline)) {
        {
            const int cur = (s - s0) & 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // ---- phase A: the next product's row pass; staging of window s + 2 (chunks in A(s,2), A(s+1,0), A(s+1,1)); stores
#ifndef FX_ABL_NOBARRIER
                if (c == 2) __syncthreads();                               // window s + 1 complete, window s no longer read
#endif
#ifndef FX_NOSB
                __builtin_amdgcn_sched_barrier(0);
#endif
                {
                    // staging chunk of this phase: (c == 2) chunk 0 of window s + 2 -> buffer cur, then the loads of window s + 3;
                    // (c == 0, 1) chunk c + 1 of window s + 1 -> buffer cur ^ 1, then the loads of window s + 2
                    const int jc = c == 2 ? 0 : c + 1, cbuf = c == 2 ? cur : cur ^ 1, snext = c == 2 ? s + 3 : s + 2;
                    const bool tvalid = s - 1 - NT >= tile0 && s > s0;
                    // slices beside the NKB window blocks: nine commit items, the finished tile's four row groups (c == 0), the loads
                    constexpr int NI = NKB >= 11 ? 1 : (NKB >= 6 ? 2 : (NKB >= 4 ? 3 : 5));     // items per slice so that nine fit
                    auto beside = [&](int kb) __attribute__((always_inline)) {
                        // the hand-off of the product phase B is about to consume (its row accumulator was read in the previous B)
                        if (NKB >= 11) {
                            // at most a dozen vector instructions beside a pair of products: conversions in blocks 0 .. 3, exchanges in 2
                            // and 4, the nine commit items in 2 .. 10, the finished tile's row groups (c == 0) in 5 .. 8, the loads last
                            if (kb == 0) split_piece(cur, c, 1);
                            if (kb == 1) split_piece(cur, c, 2);
                            if (kb == 2) { split_piece(cur, c, 3); split_piece(cur, c, 5); }
                            if (kb == 3) split_piece(cur, c, 6);
                            if (kb == 4) split_piece(cur, c, 7);
                            if (c == 0 && kb >= 5 && kb <= 8) store_group(s - 1 - NT, tvalid, kb - 5);
#ifdef FX_NO_BALANCE
                            if (kb >= 2) commit_item(cbuf, 3 * jc + (kb - 2) / 3, (kb - 2) % 3);
                            if (kb == NKB - 1) { if (jc == 0) commit_q(cbuf); issue_chunk(snext, jc); }
#else
                            // (the staging rides in phase B: beside A's two products per block there is room for the hand-off, and in
                            // A(s, 0) for the stores, but not for nine staging items on top)
                            (void)jc; (void)cbuf; (void)snext;
#endif
                        } else {
                            if (kb == 0) { split_piece(cur, c, 1); split_piece(cur, c, 2); split_piece(cur, c, 3); split_piece(cur, c, 5); split_piece(cur, c, 6); split_piece(cur, c, 7); }
#pragma unroll
                            for (int i = NI * kb; i < NI * kb + NI && i < 9; ++i) commit_item(cbuf, 3 * jc + i / 3, i % 3);
                            if (kb == NKB - 1) { if (jc == 0) commit_q(cbuf); issue_chunk(snext, jc); }
                            if (c == 0 && kb == NKB - 1) store_tile(s - 1 - NT, tvalid);
                        }
                    };
                    if (c == 2) rowpass(cur ^ 1, 0, beside); else rowpass(cur, c + 1, beside);
                }
#ifndef FX_NOSB
                __builtin_amdgcn_sched_barrier(0);
#endif
                FX_STAMP(2 * c)
                // ---- phase B: this product's column pass; the next product's hand-off; this tile's bytes
                {
                    const int nbuf = c == 2 ? cur ^ 1 : cur, nc = c == 2 ? 0 : c + 1;
                    auto beside = [&](int it) __attribute__((always_inline)) {
                        if (NKB >= 11) {
                            // S (read pieces) after triples 1, 2; E: row groups after triples 3 .. 6 (the tile that starts takes the
                            // finished tile's registers in triple NKB - 2)
                            if (it == 1) split_piece(nbuf, nc, 0, c == 2 ? s + 1 : s);
                            if (it == 2) split_piece(nbuf, nc, 4, c == 2 ? s + 1 : s);
                            if (it >= 3 && it <= 6) emit_piece(c, it - 3);
#ifndef FX_NO_BALANCE
                            // B has three products per slot and less to do beside them than A: the staging chunk of this product
                            // (c == 2: chunk 0 of window s + 2 -> buffer cur, which the barrier before A(s, 2) freed; c == 0, 1: chunk
                            // c + 1 of window s + 1 -> buffer cur ^ 1, complete before that barrier) in slots 0, 7 .. 10, its re-issue last
                            {
                                const int jc = c == 2 ? 0 : c + 1, cbuf = c == 2 ? cur : cur ^ 1, snext = c == 2 ? s + 3 : s + 2;
                                const int sl = it == 0 ? 0 : it - 6;                       // 0, 1 .. 4
                                if (it == 0 || it >= 7) {
                                    commit_item(cbuf, 3 * jc + (2 * sl) / 3, (2 * sl) % 3);
                                    if (sl < 4) commit_item(cbuf, 3 * jc + (2 * sl + 1) / 3, (2 * sl + 1) % 3);
                                }
                                if (it == NKB - 1) { if (jc == 0) commit_q(cbuf); issue_chunk(snext, jc); }
                            }
#endif
                        } else {
                            if (it == (NKB >= 5 ? 1 : 0)) { split_piece(nbuf, nc, 0, c == 2 ? s + 1 : s); split_piece(nbuf, nc, 4, c == 2 ? s + 1 : s); }
                            if (it == (NKB >= 5 ? NKB - 3 : 0)) {
#pragma unroll
                                for (int gq = 0; gq < 4; ++gq) emit_piece(c, gq);
                            }
                        }
                    };
                    colpass(c, qs, ri, beside);
                }
#ifndef FX_NOSB
                __builtin_amdgcn_sched_barrier(0);
#endif
                FX_STAMP(2 * c + 1)
            }
        }
    };
    // the first NT steps of a segment (a segment has at least NT + 1): their own copy of the body, without the column products of the
    // tiles above the segment (25 of the 55 triples of these steps per channel: 1.9 % of the matrix work of a 4K strip, 6 % of a
    // 1080p half strip's); then the rotation starts over at slot 0
#ifndef FX_NO_PEEL
#pragma unroll
    for (int j = 0; j < NT; ++j) step(s0 + j, j, j);
    for (int sb = s0 + NT; sb < s1; sb += NT) {
#else
    for (int sb = s0; sb < s1; sb += NT) {
#endif
#pragma unroll
        for (int qs = 0; qs < NT; ++qs) {
            const int s = sb + qs;
            if (s >= s1) break;
            step(s, qs, -1);
        }
    }
    store_tile(s1 - 1 - NT, s1 - 1 - NT >= tile0);
#ifdef FX_STAMPS
    if (!QUIRK && !DUMPV && vdump && blockIdx.x == 0 && tid == 0) {        // (timing-only build: vdump carries the stamp buffer)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(vdump);
        for (int i = 0; i < 6; ++i) o[i] = st_acc[i];
        o[6] = __builtin_amdgcn_s_memtime() - st_begin;
        o[7] = static_cast<unsigned long long>(s1 - s0);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The Nyquist-slot quirk (Source.cpp:420-425; derivation in mx_kernels.hpp) for the fused kernel.  With wx, wy the alternating
// weights of a reflect-101 padded line (mx_alt_weight):
//     V'[r][x]  = V[r][x] + dr (-1)^(x+pad) Srow(r),            Srow(r) = sum_x wx(x) img[r][x]                       (integers)
//     out[r][x] = colconv(V')[r][x] + dc (-1)^(r+pad) Scol(x),  Scol(x) = sum_r wy(r) V'[r][x]
//                                                                       = rowconv(Ccol)(x) + dr (-1)^(x+pad) Z
//     Ccol(x) = sum_r wy(r) img[r][x]  (integers; the row convolution and the weighted column sum commute),  Z = sum_r wy(r) Srow(r)
// fx_prepass reads the image once and leaves Srow, Ccol (whole, by integer atomics) and the parts of Z; the fused kernel adds
// qrow (-1)^x = dr (-1)^pad Srow[refl(re - PADA)][c] (-1)^x to V before the hand-off and qcol (-1)^r = dc (-1)^pad Scol (-1)^r to the
// output before the truncation (struct FxQuirk, fx_quirk_cols_tile).
#ifdef BLUR_FX_QUIRK_KERNELS   // engine.hip only: plain (non-template) kernels must live in one translation unit
// chunks at the right edge whose window (128 + 2 pada columns from 128 xc - pada) reaches past the image: they read a strip
inline int fx_right_strips(int cols, int pada)
{
    const int chunks = (cols + kFxChunk - 1) / kFxChunk;
    int n = 0;
    for (int xc = chunks - 1; xc >= fx_left_strips(pada) && kFxChunk * xc + kFxChunk + pada > cols; --xc) ++n;
    return n;
}

// strips[f][strip][row][p][3], p = 0 .. 128 + 2 pada - 1: pixel refl101(x0 - pada + p) of the row, x0 = 128 xc of the strip's chunk
// (beyond one reflection, which only zero taps read: whatever the clamped load returns).  A thread moves one group of 4 pixels: it
// lies inside the image (one 12-byte load), is the pixel-reversed copy of 4 adjacent image pixels (one 12-byte load, three
// v_perm_b32), or -- image widths that are not multiples of 4, tiny images -- straddles an edge and is gathered pixel by pixel.
// work items (fx_prepass): blocks over ceil(rows / 4) x win / 4 threads (4 rows of one group each) x strips x frames
// narrow (fx_blur_u8's strips): only the groups of fx_strip_range are written (the kernel reads the others from the image)
__device__ __forceinline__ void fx_edge_strips_body(const uint8_t* __restrict__ src, uint8_t* __restrict__ strips, int rows, int cols, int pada, int chunks, int nright,
                                                    int bx, int sidx, int f, int narrow)
{
    const int win = kFxChunk + 2 * pada, gpr = win / 4;
    const int nleft = fx_left_strips(pada);
    const int xc = sidx < nleft ? sidx : chunks - nright + sidx - nleft, x0 = kFxChunk * xc;
    int glo = 0, gn = gpr;
    if (narrow) {
        int lo, hi;
        fx_strip_range(xc, cols, pada, (gpr + 7) / 8, &lo, &hi);
        glo = 8 * lo;
        gn = min(8 * hi, gpr) - glo;
    }
    const int i = bx * 256 + threadIdx.x, rq = (rows + 3) / 4;
    if (i >= rq * gn) return;
    const int r4 = i / gn, gidx = glo + i - r4 * gn;
    const int X = x0 - pada + 4 * gidx;
    const bool mir = X < 0 || X >= cols;
    const int xs = X < 0 ? -X - 3 : (X >= cols ? 2 * cols - 5 - X : X);
    const bool fast = xs >= 0 && xs + 3 < cols;
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const uint8_t* img = src + static_cast<size_t>(f) * rows * cols * 3;
    uint8_t* sbase = strips + (static_cast<size_t>(f) * (nleft + nright) + sidx) * rows * (3 * win) + 12 * gidx;
    u3 d[4];
    if (fast) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = *reinterpret_cast<const u3*>(img + (static_cast<size_t>(min(4 * r4 + k, rows - 1)) * cols + xs) * 3);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = 4 * r4 + k;
        if (r >= rows) break;
        u3 o;
        if (fast) {
            o = d[k];
            if (mir) {       // pixels 3, 2, 1, 0 of the loaded four: bytes [9 10 11 6] [7 8 3 4] [5 0 1 2]
                o[0] = __builtin_amdgcn_perm(d[k][2], d[k][1], 0x02070605u);
                o[1] = (d[k][1] >> 24) | ((d[k][2] & 0xffu) << 8) | ((d[k][0] >> 24) << 16) | ((d[k][1] & 0xffu) << 24);
                o[2] = __builtin_amdgcn_perm(d[k][1], d[k][0], 0x02010005u);
            }
        } else {             // reaches past one reflection (tiny images): pixel by pixel
            const uint8_t* line = img + static_cast<size_t>(r) * cols * 3;
            uint32_t b[12];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const uint8_t* px = line + 3 * mx_refl(X + qd, cols);
                b[3 * qd] = px[0]; b[3 * qd + 1] = px[1]; b[3 * qd + 2] = px[2];
            }
            o[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            o[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            o[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        }
        *reinterpret_cast<u3*>(sbase + static_cast<size_t>(r) * (3 * win)) = o;
    }
}

constexpr int kFxSumRows = 32;          // image rows per sub-band (packed 16-bit column sums: 32 x 3 x 255 < 65536)
// rows per workgroup of the pre-pass: sub-bands of 32 (16 for small frames: twice the workgroups, each half as long).  Every band
// leaves a part of the column sums that every fused workgroup adds up for its own columns: tall bands where the batch still gives
// four workgroups per CU
// groups of 4 pixels per thread and row of the pre-pass: batches of 1024 gpt pixel columns, at most kFxMaxBatches of them (0: too wide)
inline int fx_groups_per_thread(int cols)
{
    const int groups = (cols + 3) / 4;
    for (int g = 1; g <= 4; g *= 2)
        if (groups <= 256 * g * kFxMaxBatches) return g;
    return 0;
}
inline int fx_band_rows(int rows, int cols, int nframes, int num_cus)
{
    if (static_cast<long long>(rows) * cols < 4000000ll) return 16;
    const int gpt = fx_groups_per_thread(cols);
    const long long nbatches = ((cols + 3) / 4 + 256 * gpt - 1) / (256 * (gpt > 0 ? gpt : 1));
    int br = kFxSumRows;
#ifndef FX_BAND_WGS
#define FX_BAND_WGS 4
#endif
    while (br < 128 && nbatches * ((rows + 2 * br - 1) / (2 * br)) * nframes >= static_cast<long long>(FX_BAND_WGS) * num_cus) br *= 2;
    return br;
}

// workgroup (band of band_rows rows, batch of 256 twelve-byte groups = 1024 pixel columns, frame), 256 threads: a thread owns one
// group (4 pixels) of every row of the band, eight rows of loads in flight; any width >= 4 (the last group of a row then overlaps the
// one before it) and any alignment (12-byte loads at byte addresses).  Exact integers:
//   srow_part[f][batch][r][c]  sum over the batch's pixels of wx(x) img[r][x][c]
//   cpart[f][band][3 x + c]    sum over the band's rows of wy(r) img[r][x][c]          (band rows 12 x groups ints apart)
//   zpart[f][band][batch][c]   sum over the band's rows of wy(r) srow_part[f][batch][r][c]       (the parts of Z)
// (Adding the parts up with atomics instead -- Srow and Ccol complete when the launch ends -- was measured: the 3 M atomic adds of
// an 8 x 4K batch cost the pre-pass 16 us, more than the consumers' few extra loads.)
// sred[row][channel][lane]: lane l of every wave adds into slot l (one conflict-free ds_add_u32 per value: a same-address atomic the
// compiler would turn into a serial loop over the lanes, and a DPP reduction costs twelve dependent instructions)
// G = groups per thread and row (1, 2, 4): a batch is 256 G groups wide, so that an image of up to 4096 G pixel columns has at most
// kFxMaxBatches batches (the fused kernel adds a row's batch parts up with that many unconditional loads)
template <int G>
__device__ __forceinline__ void fx_altsums_body(const uint8_t* __restrict__ src, int* __restrict__ srow_part, int* __restrict__ cpart, long long* __restrict__ zpart,
                                                int rows, int cols, int pad, int nbands, int nbatches, int band, int batch, int f, int (*sred)[3][64], int band_rows)
{
    const int tid = threadIdx.x;
    const uint8_t* img = src + static_cast<size_t>(f) * rows * cols * 3;
    // (a buffer resource per frame: 32-bit offsets, and nothing outside the frame can be touched)
    const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, static_cast<uint32_t>(rows) * cols * 3u, kMxRsrcWord3);
    const int groups = (cols + 3) / 4, r0 = band * band_rows, r1 = min(r0 + band_rows, rows);
    const int flip = (pad & 1) ? -1 : 1;
    // a width that is no multiple of 4 (cols >= 4): the LAST group of a row is loaded from pixel cols - 4, overlapping the group before
    // it by `over` pixels (weights 0 there, and its column sums are stored shifted by 3 over values), so that no load leaves the row
    const int over = (4 - (cols & 3)) & 3;
    int gi[G], wq[G][4];
    bool act[G], plain[G], last[G];
    uint32_t col0[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        gi[j] = (batch * G + j) * 256 + tid;
        act[j] = gi[j] < groups;
        last[j] = over != 0 && gi[j] == groups - 1;
        const int x = 4 * gi[j] - (last[j] ? over : 0);
        plain[j] = !last[j] && x > pad && x + 3 < cols - 1 - pad;         // no pixel of the group is mirrored: weights +-1 by parity
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[j][q] = (act[j] && !(last[j] && q < over)) ? mx_alt_weight(x + q, cols, pad) : 0;
        col0[j] = act[j] ? 3u * static_cast<uint32_t>(x) : 0u;
    }
    int o[G][12];                                                         // the band's column sums of the thread's bytes
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int k = 0; k < 12; ++k) o[j][k] = 0;
    long long zacc = 0;                                                   // threads 0 .. 95 = (row of the sub-band, channel)
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    for (int rs = r0; rs < r1; rs += kFxSumRows) {
        const int re = min(rs + kFxSumRows, r1);
        for (int i = tid; i < kFxSumRows * 3 * 64; i += 256) (&sred[0][0][0])[i] = 0;
        __syncthreads();
        uint32_t accp[G][6], accn[G][6];                                  // packed 16-bit sums of the rows with positive / negative wy
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < 6; ++k) accp[j][k] = accn[j][k] = 0;
#ifndef FX_PRE_RB
#define FX_PRE_RB 8
#endif
        constexpr int RB = G == 1 ? FX_PRE_RB : (G == 2 ? 4 : 2);         // rows of loads in flight (RB G loads of 12 bytes per thread)
        for (int rb = rs; rb < re; rb += RB) {
            u3 d[RB][G];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = min(rb + i, re - 1);
#pragma unroll
                for (int j = 0; j < G; ++j) d[i][j] = __builtin_amdgcn_raw_buffer_load_b96(rimg, col0[j] + static_cast<uint32_t>(r) * static_cast<uint32_t>(cols) * 3u, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = rb + i;
                if (r < re) {                                                    // uniform
                    const int wy = mx_alt_weight(r, rows, pad);
                    const uint32_t aw = static_cast<uint32_t>(wy < 0 ? -wy : wy);
                    int s[3] = { 0, 0, 0 };
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const u3 dd = d[i][j];
                        if (plain[j]) {
                            const int e0 = static_cast<int>(dd[0] ^ 0x80808080u), e1 = static_cast<int>(dd[1] ^ 0x80808080u), e2 = static_cast<int>(dd[2] ^ 0x80808080u);
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
                            s[0] += flip * t0; s[1] += flip * t1; s[2] += flip * t2;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    const int byte = 3 * q + c;
                                    s[c] += wq[j][q] * static_cast<int>((dd[byte >> 2] >> (8 * (byte & 3))) & 0xffu);
                                }
                        }
                        // column sums: bytes 0, 2 of each dword in one packed pair, bytes 1, 3 in the other; |wy| = 1, 2 or 3 (uniform)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            uint32_t lo = __builtin_amdgcn_perm(0u, dd[k], 0x0c020c00u), hi = __builtin_amdgcn_perm(0u, dd[k], 0x0c030c01u);     // bytes 0, 2 | 1, 3
                            if (aw != 1) { lo *= aw; hi *= aw; }                  // (uniform: the mirrored rows only)
                            if (wy > 0) { accp[j][2 * k] += lo; accp[j][2 * k + 1] += hi; }
                            else { accn[j][2 * k] += lo; accn[j][2 * k + 1] += hi; }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) atomicAdd(&sred[r - rs][c][tid & 63], s[c]);      // (groups right of the image: not `plain`, weights 0)
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[j][4 * k] += static_cast<int>(accp[j][2 * k] & 0xffffu) - static_cast<int>(accn[j][2 * k] & 0xffffu);
                o[j][4 * k + 2] += static_cast<int>(accp[j][2 * k] >> 16) - static_cast<int>(accn[j][2 * k] >> 16);
                o[j][4 * k + 1] += static_cast<int>(accp[j][2 * k + 1] & 0xffffu) - static_cast<int>(accn[j][2 * k + 1] & 0xffffu);
                o[j][4 * k + 3] += static_cast<int>(accp[j][2 * k + 1] >> 16) - static_cast<int>(accn[j][2 * k + 1] >> 16);
            }
        __syncthreads();
        if (tid < (re - rs) * 3) {
            const int* p64 = &sred[0][0][0] + 64 * tid;
            int v = 0;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) v += p64[(k + tid) & 63];              // (rotated: the 96 threads start on different banks)
            srow_part[((static_cast<size_t>(f) * nbatches + batch) * rows + rs) * 3 + tid] = v;
            zacc += static_cast<long long>(mx_alt_weight(rs + tid / 3, rows, pad)) * v;
        }
        __syncthreads();                                                   // sred is zeroed again / reused below
    }
    if (over != 0) {                                                       // (uniform) the last group's sums start `over` pixels into its bytes
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int v3 = o[j][k + 3], v6 = k + 6 < 12 ? o[j][k + 6] : 0, v9 = k + 9 < 12 ? o[j][k + 9] : 0;
                o[j][k] = last[j] ? (over == 1 ? v3 : (over == 2 ? v6 : v9)) : o[j][k];
            }
    }
#pragma unroll
    for (int j = 0; j < G; ++j)
        if (act[j]) {
            int4* dstp = reinterpret_cast<int4*>(cpart + (static_cast<size_t>(f) * nbands + band) * (12 * static_cast<size_t>(groups)) + 12 * gi[j]);
            dstp[0] = make_int4(o[j][0], o[j][1], o[j][2], o[j][3]);
            dstp[1] = make_int4(o[j][4], o[j][5], o[j][6], o[j][7]);
            dstp[2] = make_int4(o[j][8], o[j][9], o[j][10], o[j][11]);
        }
    long long* zs = reinterpret_cast<long long*>(&sred[0][0][0]);          // (the last sub-band's barrier has passed: sred is free)
    if (tid < 96) zs[tid] = zacc;
    __syncthreads();
    if (tid < 3) {
        long long z = 0;
        for (int k = 0; k < kFxSumRows; ++k) z += zs[3 * k + tid];
        zpart[((static_cast<size_t>(f) * nbands + band) * nbatches + batch) * 3 + tid] = z;
    }
}

// One launch for everything that has to happen before the fused kernel: the quirk's sums (n_alt = bands x batches x frames workgroups,
// none with nyquist_quirk = 0) and the edge strips (strip_blocks x nstrips x frames workgroups, last in the grid: they fill the tail).
// One kernel per G (groups per thread of the sums): the wide forms need three times the registers of G = 1
template <int G>
__global__ __launch_bounds__(256) void fx_prepass(const uint8_t* __restrict__ src, int* __restrict__ srow_part, int* __restrict__ cpart, long long* __restrict__ zpart,
                                                  uint8_t* __restrict__ strips, int rows, int cols, int pad, int pada, int nbands, int nbatches, int n_alt, int chunks,
                                                  int nright, int strip_blocks, int band_rows, int narrow)
{
    __shared__ int sred[kFxSumRows][3][64];
    int b = blockIdx.x;
    if (b < n_alt) {
        const int band = b % nbands, batch = (b / nbands) % nbatches, f = b / (nbands * nbatches);
        fx_altsums_body<G>(src, srow_part, cpart, zpart, rows, cols, pad, nbands, nbatches, band, batch, f, sred, band_rows);
    } else {
        b -= n_alt;
        const int nstrips = fx_left_strips(pada) + nright, bx = b % strip_blocks, sidx = (b / strip_blocks) % nstrips, f = b / (strip_blocks * nstrips);
        fx_edge_strips_body(src, strips, rows, cols, pada, chunks, nright, bx, sidx, f, narrow);
    }
}
#endif  // BLUR_FX_QUIRK_KERNELS

// ---- launcher ----------------------------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: remember per device which kernel sets have it (blur_multi_* drives
// several devices from one process; a process-wide flag would raise the limit on the first device only).  Devices past 63: every call.
inline bool fx_attr_needed(std::atomic<unsigned long long>& done, int& dev)
{
    dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    return ((done.load(std::memory_order_acquire) >> dev) & 1ull) == 0;
}
inline void fx_attr_mark(std::atomic<unsigned long long>& done, int dev)
{
    if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
}

struct FxEntry {
    int nkb;
    // qk: the quirk's sums (null: nyquist_quirk = 0); vdump: the row pass's float planes (test instantiation); stamps: -DFX_STAMPS builds
    hipError_t (*blur_u8)(hipStream_t, const uint8_t* src, uint8_t* dst, const void* frags, FxGeom g, int num_cus, const FxQuirk* qk, const uint8_t* strips,
                          float* vdump, unsigned long long* stamps);
};

template <int NKB> hipError_t fx_launch_u8(hipStream_t st, const uint8_t* src, uint8_t* dst, const void* frags, FxGeom g, int num_cus, const FxQuirk* qk,
                                           const uint8_t* strips, float* vdump, unsigned long long* stamps)
{
    using C = FxCfg<NKB>;
    const int chunks = (g.cols + kFxChunk - 1) / kFxChunk;
    const long long nstripes = static_cast<long long>(chunks) * g.nframes;
    if (nstripes <= 0) return hipSuccess;
    // Segments per strip of columns: a segment of t tiles takes t + NT steps (NT of run-in), and the chip runs num_cus tasks at a
    // time; take the number of segments with the shortest makespan = rounds x steps per task (one segment unless the strips
    // are too few to fill the chip: 240 strips of a 4K batch of 8 stay whole, a single 4K frame is cut in 8)
    int nseg = 1, tps = g.ntiles;
    {
        // (any number of tiles per segment: the unrolled rotation of the accumulator tiles is relative to the segment's first step.
        // Rounds 3-4 rounded it up to a multiple of NT for no reason the kernel has: 1080p, 8 frames, was cut in segments of 20 and 14
        // tiles -- 25 steps -- instead of 17 and 17 -- 22 steps)
        long long best = -1;
        for (int n = 1; n <= g.ntiles; ++n) {
            const int t = (g.ntiles + n - 1) / n, ns = (g.ntiles + t - 1) / t;
            const long long rounds = (nstripes * ns + num_cus - 1) / num_cus, span = rounds * (t + C::NT);
            if (best < 0 || span < best) { best = span; nseg = ns; tps = t; }
        }
    }
    const long long ntasks = nstripes * nseg;
    if (g.nxcd < 1) g.nxcd = 1;
    const int per_xcd = static_cast<int>((ntasks + g.nxcd - 1) / g.nxcd);
    const dim3 grid(static_cast<unsigned>(g.nxcd * per_xcd));
    static std::atomic<unsigned long long> attr_done{ 0 };
    int dev;
    if (fx_attr_needed(attr_done, dev)) {
        const void* kernels[6] = { reinterpret_cast<const void*>(fx_blur_u8<NKB, true, false>), reinterpret_cast<const void*>(fx_blur_u8<NKB, false, false>),
                                   reinterpret_cast<const void*>(fx_blur_u8<NKB, true, true>), reinterpret_cast<const void*>(fx_blur_u8<NKB, false, true>),
                                   reinterpret_cast<const void*>(fx_blur_u8<NKB, true, false, true>), reinterpret_cast<const void*>(fx_blur_u8<NKB, false, false, true>) };
        for (const void* k : kernels) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
            if (e != hipSuccess) return e;
        }
        fx_attr_mark(attr_done, dev);
    }
#define FX_LAUNCH(Q_, D_, R_)                                                                                                                              \
    hipLaunchKernelGGL((fx_blur_u8<NKB, Q_, D_, R_>), grid, dim3(256), C::LDS, st, src, dst, static_cast<const mx_half8*>(frags), g, chunks, tps, nseg, \
                       static_cast<int>(ntasks), qk ? *qk : FxQuirk{}, strips, vdump ? vdump : reinterpret_cast<float*>(stamps))
    const bool ragged = (g.cols & 3) != 0;
    if (vdump) {
        if (ragged) return hipErrorNotSupported;                 // (the row-pass dump is a test instantiation: whole quads only)
        if (qk) FX_LAUNCH(true, true, false); else FX_LAUNCH(false, true, false);
    } else if (ragged) { if (qk) FX_LAUNCH(true, false, true); else FX_LAUNCH(false, false, true); }
    else { if (qk) FX_LAUNCH(true, false, false); else FX_LAUNCH(false, false, false); }
#undef FX_LAUNCH
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
