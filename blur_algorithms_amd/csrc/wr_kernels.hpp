// wr_kernels.hpp -- "wave-resident" kernels of the 1D-tiled FFT convolution (Source.cpp:429-570).
//
// A line of N = R0 * 256 complex points (two real image lines ride in one complex line, fft_engine.hpp) is
// transformed in three stages, and only the first and the last touch LDS as a workgroup:
//   pass 0          radix-R0 decimation-in-frequency butterflies over elements j + 256 k, twiddled, written to LDS
//                   as R0 sub-blocks of 256 points;
//   middle          every sub-block is an independent 256-point transform  x multipliers x  inverse transform.  A
//                   16-lane DPP row holds one sub-block, 16 points per lane: radix-16 butterfly in registers, twiddle,
//                   lane <-> register transpose by DPP row operations (v_cndmask_b32 with row_shr / row_shl / quad_perm
//                   modifiers), radix-16 butterfly, pointwise multiply (Source.cpp:414-427, the Nyquist-slot rule is
//                   one table entry), and the same backwards.  No LDS traffic and no barrier inside: the waves of a
//                   workgroup drift apart, so one wave's LDS reads and writes hide behind the others' arithmetic;
//   inverse pass 0  conjugate twiddles, inverse radix-R0 butterflies, crop.
// pffft's in-register SIMD passes (Source.cpp:531-533,553-555) become in-register wavefront passes.
//
// The engine's transform length N need not be the reference's: in the cropped region a circular convolution of any
// length >= cols + 2 pad equals the linear convolution of the reflect-101 padded tile (Source.cpp:525-536), and the
// one place where the reference's length shows -- the Nyquist bin scaled with the DC gain, Source.cpp:420-425 -- is an
// additive term (K[0] - K[N/2]) / N_ref * (-1)^n * sum_m x[m] (-1)^m that is reproduced exactly by the multiplier of
// bin N/2 (host_math.cpp: wr_multipliers).  So N is the next multiple of 256 with a supported R0.
//
// Pass order: COLUMNS FIRST, then rows.  The two 1D passes act on different axes and commute; with the column pass
// first the awkward 3-byte pixels are only ever READ in 24-byte pieces (strip of 8 columns; over-fetch is absorbed by
// L2), the float intermediate is written as whole 64-byte sectors into per-strip contiguous blocks, and the u8 result
// leaves the row pass as whole image rows (16-byte stores).  Every HBM write is a full sector.
//
// Intermediate (per frame, per channel): [strip of 8 columns][row pair t][column 0..7][slot 0..1] floats; row r sits
// in pair t = ((r + pad) >> 1) - (pad >> 1), slot (r + pad) & 1: pairs are aligned to the PADDED row index, so that
// the two rows of a pair are neighbouring lanes (even, odd) of the column pass's last butterfly and one DPP quad_perm
// puts them side by side.  The row pass takes pair t as its complex line: re = slot 0, im = slot 1.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_engine.hpp"

namespace blur_amd {

// Diagnostic builds only (-DWR_STAMPS): shader-clock stamps around the phases of a unit, summed per WAVE and written
// behind the multiplier table (the host allocates a tail for it).  No stamp executes in a normal build.
constexpr int kWrStampSlots = 8;
constexpr int kWrStampTailFloats = 1 << 17;      // 4096 workgroup-waves x 8 slots x 8 bytes
#ifdef WR_STAMPS
#define WR_STAMP_DECL unsigned long long st_acc[kWrStampSlots] = {}; unsigned long long st_prev = __builtin_amdgcn_s_memtime()
#define WR_STAMP(i)                                                   \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                           \
        __builtin_amdgcn_sched_barrier(0);                            \
        st_acc[i] += t_ - st_prev;                                    \
        st_prev = t_;                                                 \
    } while (0)
#define WR_STAMP_FLUSH(mult, n, nwaves)                                                                                       \
    do {                                                                                                                      \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * (nwaves) + (threadIdx.x >> 6) < 2048) {                                   \
            unsigned long long* o_ = reinterpret_cast<unsigned long long*>(const_cast<float*>(mult) + (n)) +                  \
                                     (static_cast<size_t>(blockIdx.x) * (nwaves) + (threadIdx.x >> 6)) * kWrStampSlots;       \
            for (int i_ = 0; i_ < kWrStampSlots; ++i_) o_[i_] = st_acc[i_];                                                   \
        }                                                                                                                     \
    } while (0)
#else
#define WR_STAMP_DECL do { } while (0)
#define WR_STAMP(i) do { } while (0)
#define WR_STAMP_FLUSH(mult, n, nwaves) do { } while (0)
#endif

#ifdef WR_ABL_NOMID
#define WR_MID_ON(on_) ((on_) && blockIdx.x == 0x7fffffff)
#else
#define WR_MID_ON(on_) (on_)
#endif

// Makes a per-thread value opaque to the optimiser at this point of the loop body.  Without it everything that follows
// from a loop-invariant per-thread index (sixteen load offsets, fifteen LDS table addresses, ...) is hoisted out of the
// persistent unit loop, kept in registers for the whole kernel and spilled (fast_kernels.hpp: 168 -> 102 VGPRs).
template <class T_> __device__ __forceinline__ T_ wr_opaque(T_ v_)
{
    asm volatile("" : "+v"(v_));
    return v_;
}

constexpr int kWrS = 256;          // points of a wave-resident sub-transform: 16 lanes x 16 registers
constexpr int kWrSB = kWrS + 8;    // LDS stride of a sub-block (complex elements): the spare 64 bytes put the four
                                   // sub-blocks a wave works on (lane & 3) on disjoint banks when it reads them

// ---- lane <-> register transpose ---------------------------------------------------------------------------
// The 16 lanes that hold one sub-block are the lanes of a wave with equal (lane & 3): lane bits 2..5 number them
// (lambda = (lane >> 2) & 15), one lane out of every quad.  Register bit S is exchanged with lambda bit S:
//   a = v[r] (bit S of r clear), b = v[r | 1 << S]:   a'[l] = hi(l) ? b[partner(l)] : a[l],   b'[l] = hi(l) ? b[l] : a[partner(l)]
// with hi(l) = lambda bit S of lane l and partner(l) = the lane whose lambda differs in bit S.
//   lambda bit 0 = lane bit 2: DPP row_shr:4 / row_shl:4, the select is the DPP bank mask (a bank = 4 lanes)
//   lambda bit 1 = lane bit 3: DPP row_shr:8 / row_shl:8, bank mask
//   lambda bit 2 = lane bit 4: v_permlane16_swap_b32 (odd rows of a <-> even rows of b, one instruction)
//   lambda bit 3 = lane bit 5: v_permlane32_swap_b32 (upper half of a <-> lower half of b)
// x of another lane of the same 16-lane DPP row (dpp_ctrl: quad_perm 0x00-0xFF, row_shl:n 0x100+n, row_shr:n 0x110+n)
template <int CTRL> __device__ __forceinline__ float wr_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

template <int S> __device__ __forceinline__ void wr_xchg(float& a, float& b)
{
    const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
    if constexpr (S == 0) {
        const int na = __builtin_amdgcn_update_dpp(ia, ib, 0x114, 0xf, 0xA, false);   // row_shr:4 into banks 1,3
        const int nb = __builtin_amdgcn_update_dpp(ib, ia, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0,2
        a = __builtin_bit_cast(float, na);
        b = __builtin_bit_cast(float, nb);
    } else if constexpr (S == 1) {
        const int na = __builtin_amdgcn_update_dpp(ia, ib, 0x118, 0xf, 0xC, false);   // row_shr:8 into banks 2,3
        const int nb = __builtin_amdgcn_update_dpp(ib, ia, 0x108, 0xf, 0x3, false);   // row_shl:8 into banks 0,1
        a = __builtin_bit_cast(float, na);
        b = __builtin_bit_cast(float, nb);
    } else {
        static_assert(S == 0 || S == 1, "lambda bits 2 and 3: wr_swap4");
    }
}

// lambda bits 2 and 3: four register pairs per asm statement.  (The builtins __builtin_amdgcn_permlane16_swap /
// permlane32_swap return both new values, but ROCm 7.2's optimiser folds the second result into the first.)
// A VALU write of a register needs two wait states before a permlane swap reads it: the leading s_nop covers whatever
// precedes the statement; the four swaps touch different registers; a VALU read right after a swap is safe.
template <int S> __device__ __forceinline__ void wr_swap4(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2, float& a3, float& b3)
{
    if constexpr (S == 2)
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
            : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
    else
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
            : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
}

template <int S> __device__ __forceinline__ void wr_xchg_step(float2 (&v)[16])
{
    constexpr int D = 1 << S;
    if constexpr (S < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r & D) == 0) {
                wr_xchg<S>(v[r].x, v[r | D].x);
                wr_xchg<S>(v[r].y, v[r | D].y);
            }
        }
    } else {
        // the eight registers with bit S clear are {0..3} + {0, O}: O = 8 for S = 2, 4 for S = 3; four at a time
        constexpr int O = S == 2 ? 8 : 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = h * O;
            wr_swap4<S>(v[b].x, v[b | D].x, v[b + 1].x, v[(b + 1) | D].x, v[b + 2].x, v[(b + 2) | D].x, v[b + 3].x, v[(b + 3) | D].x);
            wr_swap4<S>(v[b].y, v[b | D].y, v[b + 1].y, v[(b + 1) | D].y, v[b + 2].y, v[(b + 2) | D].y, v[b + 3].y, v[(b + 3) | D].y);
        }
    }
}

// v[r] of lane lambda  <->  v[lambda] of lane r, among the 16 lanes of a sub-block (tools/wr_model.py: transpose16)
// (ablation builds, timing only, results are wrong: -DWR_ABL_NOTR no transposes, -DWR_ABL_NOSWAP no permlane swaps,
//  -DWR_ABL_NODPP no DPP steps, -DWR_ABL_NOMID no middle section at all)
__device__ __forceinline__ void wr_transpose16(float2 (&v)[16])
{
#ifndef WR_ABL_NOTR
#ifndef WR_ABL_NODPP
    wr_xchg_step<0>(v);
    wr_xchg_step<1>(v);
#endif
#ifndef WR_ABL_NOSWAP
    wr_xchg_step<2>(v);
    wr_xchg_step<3>(v);
#endif
#endif
}

// per-thread constants of the middle section: tw[k] = exp(-2 pi i (lane & 15) k / 256), mm[rho] = multiplier of
// frequency q + R0 (lane16 + 16 rho) of the line (natural order table)
// TWL = false: tw in registers (30 VGPRs); true: read where used from a 2 KB LDS table [k][lambda] (the row pass, whose
// register budget goes to the prefetched next unit)
template <bool TWL> struct WrMid {
    float2 tw[TWL ? 1 : 16];
    const float2* twl;      // TWL: LDS table, entry k * 16 + lambda = exp(-2 pi i lambda k / 256)
    float mm[16];
    __device__ __forceinline__ float2 w(int k, int lane16) const
    {
        if constexpr (TWL) return twl[k * 16 + lane16];
        else return tw[k];
    }
};
constexpr size_t kWrTwlBytes = 16 * 16 * sizeof(float2);

// fills the LDS twiddle table of the TWL variant (all threads; a barrier must follow)
__device__ __forceinline__ void wr_twl_fill(float2* twl, const float2* __restrict__ w256, int tid, int nthreads)
{
    for (int i = tid; i < 256; i += nthreads) twl[i] = w256[((i >> 4) * (i & 15)) & 255];
}

template <int R0, bool TWL> __device__ __forceinline__ void wr_mid_load(WrMid<TWL>& w, int lane16, int q, const float2* __restrict__ w256, const float* __restrict__ mult, const float2* twl = nullptr)
{
    if constexpr (!TWL) {
#pragma unroll
        for (int k = 1; k < 16; ++k) w.tw[k] = w256[(lane16 * k) & 255];
        w.tw[0] = make_float2(1.f, 0.f);
    }
    w.twl = twl;
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) w.mm[rho] = mult[q + R0 * (lane16 + 16 * rho)];
}

// forward 256-point transform, pointwise multiply, inverse transform of the sub-block at `sb` (LDS), in place.
// Lane lambda (= lane16) of the sub-block's 16 lanes holds elements lambda + 16 k.
// hook(0..3) is called at four points of the first half of the section: the kernels use it to trickle the next unit's
// global loads into the memory system one at a time instead of in a burst (a CU-wide burst of ~1500 scattered line
// requests stalls the issuing waves for as long as HBM needs to serve it; measured with -DWR_STAMPS).
struct WrNoHook {
    static constexpr bool pinned = false;
    __device__ __forceinline__ void operator()(int, float) const {}
};
// a hook wrapped in scheduling barriers: its loads stay where they are written instead of being hoisted to the top
template <class F> struct WrPinned {
    static constexpr bool pinned = true;
    F f;
    // `dep`: a value the section has just computed; the hook is ordered behind it (IR passes move pure arithmetic freely)
    __device__ __forceinline__ void operator()(int q, float dep) const
    {
        asm volatile("" ::"v"(dep));
        __builtin_amdgcn_sched_barrier(0);
        f(q);
        __builtin_amdgcn_sched_barrier(0);
    }
};
template <class F> __device__ __forceinline__ WrPinned<F> wr_pinned(F f) { return WrPinned<F>{ f }; }
template <bool TWL, class Hook = WrNoHook> __device__ __forceinline__ void wr_middle(float2* sb, const WrMid<TWL>& w, int lane16, Hook hook = Hook())
{
    float2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sb[16 * k + lane16];
    hook(0, v[15].y);
    Bfly<16, false>::run(v);
    hook(1, v[15].y);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w.w(k, lane16));
    wr_transpose16(v);
    hook(2, v[15].y);
    Bfly<16, false>::run(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = cscale(v[r], w.mm[r]);
    hook(3, v[15].y);
    Bfly<16, true>::run(v);
#pragma unroll
    for (int r = 1; r < 16; ++r) v[r] = cmulc(v[r], w.w(r, lane16));
    wr_transpose16(v);
    Bfly<16, true>::run(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) sb[16 * k + lane16] = v[k];
}

// reflect-101 source index of padded position p (Source.cpp:525-529); -1: beyond the padded tile (zero)
__device__ __forceinline__ int wr_reflect(int p, int pad, int len)
{
    int i = p - pad;
    i = i < 0 ? -i : i;
    const int m = 2 * (len - 1) - i;
    const int r = i < len ? i : m;
    return p < len + 2 * pad ? r : -1;
}

// pass 0, second half: butterfly, twiddle w_N^(j q), store element j of sub-block q
template <int R0> __device__ __forceinline__ void wr_p0_store(float2 (&v)[R0], float2* line, const float2* tw0, int j)
{
    Bfly<R0, false>::run(v);
    line[j] = v[0];
#pragma unroll
    for (int q = 1; q < R0; ++q) line[q * kWrSB + j] = cmul(v[q], tw0[(q - 1) * kWrS + j]);
}
// inverse pass 0, first half: load, conjugate twiddle, inverse butterfly: v[k] = element j + 256 k of the convolved line
template <int R0> __device__ __forceinline__ void wr_ip0_load(float2 (&v)[R0], const float2* line, const float2* tw0, int j)
{
    v[0] = line[j];
#pragma unroll
    for (int q = 1; q < R0; ++q) {
        v[q] = cmulc(line[q * kWrSB + j], tw0[(q - 1) * kWrS + j]);
        // at most eight twiddles in flight: read all up front, a radix-16 pass holds 30 registers of them (row pass: spills)
        if (R0 > 10 && q % 8 == 0) __builtin_amdgcn_sched_barrier(0);
    }
    Bfly<R0, true>::run(v);
}

// persistent workgroups, XCD-aware: workgroups b, b + 8, ... share an XCD (round-robin dispatch: speed only).  Each XCD
// gets one contiguous run of units and its workgroups walk it interleaved, so that units whose data share 128-byte
// lines are in flight together and meet in that XCD's L2.
struct WrWalk {
    int begin, end, step;
};
__device__ __forceinline__ WrWalk wr_walk(int nunits)
{
    const int nx = gridDim.x < 8 ? static_cast<int>(gridDim.x) : 8;
    const int xcd = blockIdx.x % nx, rank = blockIdx.x / nx;
    const int wg_in_xcd = (static_cast<int>(gridDim.x) - xcd + nx - 1) / nx;
    const int b = static_cast<int>(static_cast<long long>(xcd) * nunits / nx);
    const int e = static_cast<int>(static_cast<long long>(xcd + 1) * nunits / nx);
    return WrWalk{ b + rank, e, wg_in_xcd };
}

template <int R0, int C> __host__ __device__ constexpr size_t wr_lines_bytes() { return static_cast<size_t>(C) * R0 * kWrSB * sizeof(float2); }
// column pass: lines 4 complex elements further apart, so that the four lines a 16-lane group touches in pass 0 (lane
// bits 1-2 select the line) fall on different banks
template <int R0> __host__ __device__ constexpr int wr_col_line_stride() { return R0 * kWrSB + 4; }
template <int R0> __host__ __device__ constexpr size_t wr_tw0_bytes() { return static_cast<size_t>(R0 - 1) * kWrS * sizeof(float2); }

// ======================================================================================
// column pass (first): u8 pixels -> float intermediate
// ======================================================================================
// One workgroup per CU, persistent over (frame, strip of 2C = 8 columns).  The strip's raw bytes (24 per image row)
// arrive by 8-byte loads issued into registers one strip ahead, are parked in an LDS byte stage and serve the three
// channels one after the other (deinterleave_BGR, Utils.hpp:159-184, is the byte pick).  A task = (strip, channel):
// pass 0 over the C complex lines (line l = columns 2l, 2l+1; reflect-101 along the column, Source.cpp:549-551), middle,
// inverse pass 0, and the cropped rows (Source.cpp:558) leave as 8-byte stores: the even lane of a lane pair holds row
// p, the odd lane row p + 1 of the same two columns; one quad_perm exchange gives each lane one column with both rows.
//
// Two phases and two barriers per task: butterfly j of line l reads and writes only elements (l, q, j), so the inverse
// pass 0 of task t-1 and the pass 0 of task t are ONE phase (in place, no barrier between them; their arithmetic and
// LDS traffic interleave), and the middle of task t is the other.
// rows of the LDS byte stage are one dword further apart than their 6C bytes: with 24-byte rows the 64 lanes of a pass-0
// byte read (16 rows x 4 lines) pile three deep on the 32 banks; 7 dwords per row are coprime to the bank count
template <int C> __host__ __device__ constexpr int wr_col_stage_row() { return C == 2 ? 12 : 2 * C * 3 + 4; }     // (C = 2: 3 dwords are odd already)
template <int R0, int C> __host__ __device__ constexpr size_t wr_col_lds(int rows)
{
    return static_cast<size_t>(C) * wr_col_line_stride<R0>() * sizeof(float2) + wr_tw0_bytes<R0>() + ((static_cast<size_t>(rows) + 1) * wr_col_stage_row<C>() + 15) / 16 * 16;
}

// Tiled images (engine.hip: run_wr_tiled): an image whose lines are longer than the longest wave-resident transform is cut into
// bands of rows (column pass) and tiles of columns (row pass) with the kernel's reach of real pixels either side -- a band or tile
// in the middle of the image is a circular convolution without any border (pad = 0 for the kernel; the contaminated ends are not
// kept) -- and the Nyquist-slot quirk of pffft_() (Source.cpp:420-425), which is a property of the WHOLE padded line, enters as
// rank-one terms made from the integer sums of fx_prepass:
//   intermediate W'[r][x] = colconv(img)[r][x] + (-1)^(r + pad) e(x),   e(x) = dc Ccol(x)              (WrColTerm)
//   out[r][x] = rowconv(W')[r][x] + (-1)^(x + pad) h(r),                h(r) = dr (colconv(Srow)(r) + dc (-1)^(r + pad) Z)   (WrRowTile)
// with multiplier tables WITHOUT the quirk (blur_opts.nyquist_quirk = 0 is the same without the terms).
struct WrColTerm {
    const float* e = nullptr;      // [frame][channel][pitch]: e(x) per column (nullptr: no term)
    int pitch = 0;                 // floats per (frame, channel): the image width rounded up to 8
    float sign = 1.f;              // (-1)^(r + pad) of slot 0 of a row pair of this call
    int band_step = 0;             // > 0: the call's "frames" are bands of ONE image, band_step (even) image rows apart
};
// what a row-pass call on a band / tile needs beyond the plain call (defaults = the whole image)
struct WrRowTile {
    const float* h = nullptr;      // [frame][channel][rows_full]: h(r) per image row (nullptr: no term)
    int rows_full = 0;             // image rows (pitch of h, rows of the output frame); 0: the call's rows
    int row_base = 0;              // image row of the call's row 0
    int vr0 = 0, vr1 = 0;          // image rows [vr0, vr1) are written (vr1 = 0: every row of the call)
    int ypar = -1;                 // parity of the row pairs ((row + pad of the COLUMN call) & 1 = slot); -1: pad & 1 of this call
    int dst_pitch = 0;             // bytes between output rows (0: the call's cols * 3)
    int dst_x0 = 0;                // image column of the call's column 0
    int vc0 = 0, vc1 = 0;          // columns [vc0, vc1) of the call are written (vc1 = 0: all); vc0 a multiple of 16
    int xpar = 0;                  // parity of (image column of the call's column 0 + the kernel's half width): sign of h's term
    int plane_strips = 0;          // strips of 8 columns per channel plane of the intermediate (0: the call's)
    int t0 = 0, tn = 0;            // row pairs [t0, t0 + tn) of a frame are transformed (tn = 0: all): the others hold no row that is written
    int g = 8;                     // columns per strip of the intermediate (WrEntry::g of the column kernel that wrote it)
    int band_step = 0;             // > 0: the call's "frames" are bands of ONE image, band_step image rows apart (row_base, vr0, vr1 of band 0)
    int vr_max = 0;                // band_step > 0: no image row from vr_max on is written (the last band may reach into the next span's rows)
};

// ELO / EHI: the first ELO and the last EHI rounds k of pass 0 may touch the reflected borders or the zero tail (index
// arithmetic per element); the rounds between are interior for every thread (checked by the launcher).
template <int R0, int C, int T, int ELO, int EHI>
__global__ __launch_bounds__(T) void wr_colpass_u8(const uint8_t* __restrict__ src, float* __restrict__ inter,
                                                   int rows, int cols, int pad, int npairs, int nstrips, int nunits, int aligned8,
                                                   const float2* __restrict__ w256, const float2* __restrict__ tw0g, const float* __restrict__ mult,
                                                   WrColTerm term)
{
    constexpr int CH = 3, G = 2 * C, RB = G * CH;
    constexpr int RS = wr_col_stage_row<C>();        // bytes between rows of the LDS byte stage
    constexpr int NSB = C * R0;                       // sub-blocks per task
    constexpr int LS = wr_col_line_stride<R0>();      // complex elements between lines
    static_assert(C == 4 || C == 2, "lane numbering of pass 0: bit 0 = row parity, bits 1 .. log2 C = line, the bits above = row pair");
    static_assert(NSB * 16 <= T && T % 64 == 0, "the middle section is one round");
    static_assert(RB % 8 == 0 || RB == 12, "a strip row is a whole number of 8-byte pieces, or one 12-byte piece (C = 2)");
    constexpr int LB = C == 4 ? 2 : 1;                // log2 C
    constexpr int total0 = C * kWrS, IT0 = (total0 + T - 1) / T;
    constexpr int N = R0 * kWrS;
    constexpr int NE = ELO + EHI > 0 ? ELO + EHI : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lines = reinterpret_cast<float2*>(smem);
    float2* tw0 = lines + C * LS;
    uint8_t* stage = reinterpret_cast<uint8_t*>(tw0 + (R0 - 1) * kWrS);      // [rows + 1][RS]; row `rows` stays zero

    const int tid = threadIdx.x;
    for (int i = tid; i < (R0 - 1) * kWrS; i += T) tw0[i] = tw0g[i];
    for (int i = tid; i < RS; i += T) stage[rows * RS + i] = 0;
    // middle section: sub-block sbi = l * R0 + q at l * LS + q * kWrSB
    const int lane16 = (tid >> 2) & 15, sbi = (tid >> 6) * 4 + (tid & 3);   // wr_transpose16's lane numbering
    const bool mid_on = sbi < NSB;
    const int sb_off = mid_on ? (sbi / R0) * LS + (sbi % R0) * kWrSB : 0;
    WrMid<false> wm;
    wr_mid_load<R0>(wm, lane16, mid_on ? sbi % R0 : 0, w256, mult);

    // raw strip rows in registers, 8-byte pieces, requested a few at a time over the three tasks before they are needed
    constexpr int PB = RB % 8 == 0 ? 8 : 12, PD = PB / 4;      // bytes / dwords per piece
    constexpr int PIECES = RB / PB;
    constexpr int KU = (PIECES * N + T - 1) / T;      // rows < N
    constexpr int KP = (KU + 2) / 3;                  // pieces per task
    typedef unsigned int wr_piece __attribute__((ext_vector_type(PD)));
    wr_piece pfu[KU];
    const size_t frame_bytes = static_cast<size_t>(term.band_step > 0 ? term.band_step : rows) * cols * CH;     // bands of one image overlap in memory
    const int npiece = rows * PIECES;
    // 1: a whole strip; 2: the last strip of an image whose width is no multiple of G -- its pieces reach into the next row: the same
    // loads, the bytes right of the image masked when they are committed, the LAST row (whose pieces would leave the frame) byte by
    // byte; 0: byte by byte altogether.  (Until round 4 every ragged strip went byte by byte: rows x 12 dependent byte loads by one
    // workgroup at the tail of the launch -- 131 us against 88 for the 2050-pixel-wide image of the reference's sweep, half of whose
    // sizes have widths = 2 mod 4.)
    auto strip_kind = [&](int uu) { const int ss = uu % nstrips; return aligned8 == 0 ? 0 : ((ss + 1) * G <= cols ? 1 : (rows > 1 ? 2 : 0)); };
    // pieces [k0, k1) of strip uu
    auto issue_strip = [&](int uu, int k0, int k1) {
        const int kind = strip_kind(uu);
        if (kind == 0) return;
        const int ff = uu / nstrips, ss = uu - ff * nstrips;
        const uint8_t* base = src + static_cast<size_t>(ff) * frame_bytes + static_cast<size_t>(ss) * RB;
        const int t0 = wr_opaque(tid);
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (k < k0 || k >= k1) continue;
            int idx = t0 + T * k;
            idx = idx < npiece ? idx : npiece - 1;              // unconditional loads: all in flight together
            int r = idx / PIECES;
            const int d = idx - r * PIECES;
            r = (kind == 2 && r == rows - 1) ? r - 1 : r;       // (the last row of a ragged strip is fetched byte by byte: commit_strip)
#ifdef WR_ABL_NOLOAD
            pfu[k] = wr_piece(static_cast<unsigned>(r + d));
            (void)base;
#else
            pfu[k] = *reinterpret_cast<const wr_piece*>(base + static_cast<size_t>(r) * cols * CH + PB * d);
#endif
        }
    };
    auto claim_strip = [&]() {
#pragma unroll
        for (int k = 0; k < KU; ++k) {
#pragma unroll
            for (int d = 0; d < PD; ++d) asm volatile("" ::"v"(pfu[k][d]));
        }
    };
    auto commit_strip = [&](int uu) {
        const int kind = strip_kind(uu);
        if (kind == 1) {
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int idx = tid + T * k;
                if (idx < npiece) {
                    const int r = idx / PIECES, d = idx - r * PIECES;
                    unsigned* o = reinterpret_cast<unsigned*>(stage + r * RS + PB * d);      // rows are 4-byte aligned only
#pragma unroll
                    for (int w = 0; w < PD; ++w) o[w] = pfu[k][w];
                }
            }
        } else if (kind == 2) {
            const int ff = uu / nstrips, ss = uu - ff * nstrips;
            const int vb = (cols - ss * G) * CH;                 // bytes of a strip row that lie inside the image
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int idx = tid + T * k;
                const int r = idx / PIECES, d = idx - r * PIECES;
                if (idx < npiece && r < rows - 1) {
                    unsigned* o = reinterpret_cast<unsigned*>(stage + r * RS + PB * d);
                    const int nv = min(max(vb - PB * d, 0), PB);
#pragma unroll
                    for (int w = 0; w < PD; ++w) {
                        const int keep = min(max(nv - 4 * w, 0), 4);
                        o[w] = pfu[k][w] & (keep >= 4 ? 0xffffffffu : ((1u << (8 * keep)) - 1u));
                    }
                }
            }
            const uint8_t* lastrow = src + static_cast<size_t>(ff) * frame_bytes + static_cast<size_t>(rows - 1) * cols * CH;
            for (int b = tid; b < RB; b += T) {
                const int col = ss * G + b / CH;
                stage[(rows - 1) * RS + b] = col < cols ? lastrow[static_cast<size_t>(col) * CH + (b % CH)] : static_cast<uint8_t>(0);
            }
        } else {
            // ragged last strip or rows that are not 8-byte aligned: byte loads, columns beyond the image are zero
            const int ff = uu / nstrips, ss = uu - ff * nstrips;
            const uint8_t* base = src + static_cast<size_t>(ff) * frame_bytes;
            for (int idx = tid; idx < rows * RB; idx += T) {
                const int r = idx / RB, b = idx - r * RB;
                const int col = ss * G + b / CH;
                stage[r * RS + b] = col < cols ? base[(static_cast<size_t>(r) * cols + col) * CH + (b % CH)] : static_cast<uint8_t>(0);
            }
        }
    };

    // Pass-0 butterfly g (0 .. C*256-1) -> (line l, butterfly j): inside a wave, lane bit 0 = j & 1 (the two rows of a
    // pair), bits 1-2 = l, bits 3-5 = (j >> 1) & 7.  The eight lanes of a row pair then hold the pair's whole 64-byte
    // record, and a wave's store covers eight consecutive records: 512 contiguous bytes.
    // (C = 2: bit 1 = l, bits 2-5 = (j >> 1) & 15: the four lanes of a row pair hold its 32-byte record, a wave's store covers sixteen)
    auto line_of = [](int g) { return (g >> 1) & (C - 1); };
    auto bfly_of = [](int g) { return ((g >> 6) << (6 - LB)) | (((g >> (1 + LB)) & ((1 << (5 - LB)) - 1)) << 1) | (g & 1); };
    // per-thread, per-round constants of the two pass-0 halves (loop invariant: a handful of registers)
    //   a_in[it]   byte address inside the stage of element (l, j - pad): interior rounds add k * 256 * RS as an immediate
    //   a_e[it][e] the same for the edge rounds (reflected row, or the zero row)
    //   o_out[it]  float offset inside the strip's output block of pair ((j >> 1) - (pad >> 1)), this lane's column
    int a_in[IT0], a_e[IT0][NE], o_out[IT0];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int g = tid + T * it, l = line_of(g), j = bfly_of(g) & 255;
        a_in[it] = (j - pad) * RS + (2 * l) * CH;
#pragma unroll
        for (int e = 0; e < ELO + EHI; ++e) {
            const int k = e < ELO ? e : R0 - EHI + (e - ELO);
            int r = wr_reflect(j + k * kWrS, pad, rows);
            r = r < 0 ? rows : r;
            a_e[it][e] = r * RS + (2 * l) * CH;
        }
        o_out[it] = ((j >> 1) - (pad >> 1)) * (2 * G) + 2 * (2 * l + (j & 1));
    }

    WR_STAMP_DECL;
    const WrWalk walk = wr_walk(nunits);
    if (walk.begin < walk.end) {
        issue_strip(walk.begin, 0, KU);
        claim_strip();
        commit_strip(walk.begin);
        if (walk.begin + walk.step < walk.end) issue_strip(walk.begin + walk.step, 0, KP);      // as the last task of a strip before would have
    }
    __syncthreads();
    WR_STAMP(7);      // prologue: tables, first strip
    const size_t plane = static_cast<size_t>(nstrips) * npairs * (2 * G);     // floats per channel

    // the term of the task whose inverse pass 0 is due: this lane's column of (strip, channel), slot 0 gets + , slot 1 -
    float tv[IT0];
#pragma unroll
    for (int it = 0; it < IT0; ++it) tv[it] = 0.f;
    auto load_term = [&](int ff, int strip_, int ch_) {
        if (term.e == nullptr) return;                   // uniform
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int g = tid + T * it;
            const int x = strip_ * G + 2 * line_of(g) + (bfly_of(g) & 1);
            tv[it] = term.sign * term.e[(static_cast<size_t>(term.band_step > 0 ? 0 : ff) * CH + ch_) * term.pitch + (x < term.pitch ? x : 0)];
        }
    };
    // inverse pass 0 of the task whose output block is `out_blk`: LDS -> butterfly -> lane-pair exchange -> global
    auto inverse_pass0 = [&](float* out_blk) {
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int g = wr_opaque(tid + T * it);
            if (IT0 * T == total0 || g < total0) {
                const int l = line_of(g), j = bfly_of(g);
                float2 v[R0];
                wr_ip0_load<R0>(v, lines + l * LS, tw0, j);
                const bool odd = (j & 1) != 0;
                float* const ob = out_blk + wr_opaque(o_out[it]);
#pragma unroll
                for (int k = 0; k < R0; ++k) {
                    // even lane: row p (slot 0) of columns 2l, 2l+1; odd lane: row p + 1 (slot 1) of the same columns
                    const float px = wr_dpp<0xB1>(v[k].x), py = wr_dpp<0xB1>(v[k].y);
                    float2 o = odd ? make_float2(py, v[k].y) : make_float2(v[k].x, px);
                    o.x += tv[it];                        // (tiled images: the column pass's Nyquist-slot term e(x) (-1)^(r + pad), WrColTerm)
                    o.y -= tv[it];
                    bool ok = true;
                    if (k < ELO || k >= R0 - EHI) {
                        const int t = ((j + k * kWrS) >> 1) - (pad >> 1);
                        ok = t >= 0 && t < npairs;
                    }
#ifdef WR_ABL_NOSTORE
                    asm volatile("" ::"v"(o.x), "v"(o.y), "v"(ob));
                    if (false)
#else
                    if (ok)
#endif
                        *reinterpret_cast<float2*>(ob + k * (kWrS / 2) * (2 * G)) = o;
                }
            }
        }
    };

    float* prev_blk = nullptr;
    for (int u = walk.begin; u < walk.end; u += walk.step) {
        const int f = u / nstrips, strip = u - f * nstrips;
        for (int ch = 0; ch < CH; ++ch) {
            // ---- phase A: inverse pass 0 of the previous task, pass 0 of this one (in place, same elements per thread)
            if (prev_blk) inverse_pass0(prev_blk);
            WR_STAMP(4);      // inverse pass 0 + stores
            const uint8_t* const stage_ch = stage + ch;
#pragma unroll
            for (int it = 0; it < IT0; ++it) {
                const int g = wr_opaque(tid + T * it);
                if (IT0 * T == total0 || g < total0) {
                    const int l = line_of(g), j = bfly_of(g);
                    float2 v[R0];
                    const int a_in_ = wr_opaque(a_in[it]);
#pragma unroll
                    for (int k = 0; k < R0; ++k) {
                        const uint8_t* s;
                        if (k >= ELO && k < R0 - EHI) s = stage_ch + a_in_ + k * (kWrS * RS);      // the whole round is interior
                        else s = stage_ch + a_e[it][k < ELO ? k : ELO + (k - (R0 - EHI))];
                        v[k] = make_float2(static_cast<float>(s[0]), static_cast<float>(s[CH]));
                    }
                    wr_p0_store<R0>(v, lines + l * LS, tw0, j);
                }
            }
            WR_STAMP(0);      // pass 0
            __syncthreads();
            WR_STAMP(1);      // barrier after phase A
            // ---- phase B: the middle (wave-resident, no barrier inside).  The byte stage changes hands on the strip's
            // last channel (every byte of this strip has been read: park the next strip); the strip after that is
            // requested piece by piece from inside the middle sections of three tasks.
            const int un = u + walk.step;
            if (ch == CH - 1 && un < walk.end) commit_strip(un);
            WR_STAMP(6);      // commit of the next strip (last channel only)
            // last channel: the first third of the strip after the next; channel 0: the other two thirds of the next
            // strip; channel 1: nothing, and the claim at its end finds loads that are a whole task old
            const int target = ch == CH - 1 ? un + walk.step : un;
            constexpr int KQ = (KP + 3) / 4, KQ2 = (2 * KP + 3) / 4;          // pieces per hook call
            auto hook = wr_pinned([&](int q) {
                if (target < walk.end) {
                    // compile-time register ranges (the piece registers must never be indexed dynamically)
                    if (ch == CH - 1) issue_strip(target, q * KQ, (q + 1) * KQ < KP ? (q + 1) * KQ : KP);
                    else if (ch == 0) issue_strip(target, KP + q * KQ2, KP + (q + 1) * KQ2 < KU ? KP + (q + 1) * KQ2 : KU);
                }
            });
            if (WR_MID_ON(mid_on)) wr_middle(lines + wr_opaque(sb_off), wm, wr_opaque(lane16), hook);
            else { hook(0, 0.f); hook(1, 0.f); hook(2, 0.f); hook(3, 0.f); }
            WR_STAMP(2);      // middle
            if (ch == 1) claim_strip();      // the strip that is parked next; before the next phase's stores (vmcnt retires in order)
            __syncthreads();
            WR_STAMP(3);      // barrier after phase B (+ claim)
            prev_blk = inter + (static_cast<size_t>(f) * CH + ch) * plane + static_cast<size_t>(strip) * npairs * (2 * G);
            load_term(f, strip, ch);
        }
    }
    if (prev_blk) inverse_pass0(prev_blk);
    WR_STAMP(4);
    WR_STAMP_FLUSH(mult, N, T / 64);
}

// ======================================================================================
// row pass (second): float intermediate -> u8 pixels
// ======================================================================================
// One workgroup of 3 * 256 threads per CU, persistent over (frame, row pair); the three channel lines of the pair are
// transformed together.  Pass 0 takes its input straight from global memory: thread (c, j) owns columns
// reflect(j + 256 k - pad) (Source.cpp:525-529), one 8-byte load each (slot 0, slot 1 = re, im), requested one unit
// ahead into registers.  The last butterfly applies interleave_BGR's "+0.5f, truncate" (Utils.hpp:189,204-206) into an
// LDS byte stage that leaves as two whole image rows.  Two phases per unit, as in the column pass: {inverse pass 0 of
// unit u-1, pass 0 of unit u} and {write-out of unit u-1, middle of unit u}.
template <int R0> __host__ __device__ constexpr size_t wr_row_lds(int cols)
{
    return wr_lines_bytes<R0, 3>() + wr_tw0_bytes<R0>() + kWrTwlBytes + 2 * ((static_cast<size_t>(cols) * 3 + 15) / 16 * 16);
}

// G: columns per strip of the intermediate the column kernel wrote (8: C = 4 lines per task; 4: the C = 2 kernels for long columns)
template <int R0, int T, int ELO, int EHI, int G = 8>
__global__ __launch_bounds__(T) void wr_rowpass_u8(const float* __restrict__ inter, uint8_t* __restrict__ dst,
                                                   int rows, int cols, int pad, int npairs, int nstrips, int nunits, int aligned16,
                                                   const float2* __restrict__ w256, const float2* __restrict__ tw0g, const float* __restrict__ mult,
                                                   WrRowTile tile)
{
    constexpr int CH = 3, C = 3, LG = G == 8 ? 3 : 2;
    static_assert(G == 8 || G == 4, "strips of 8 or 4 columns");
    constexpr int NSB = C * R0;
    constexpr int NE = ELO + EHI > 0 ? ELO + EHI : 1;
    static_assert(T == C * kWrS, "pass 0: one butterfly per thread");
    static_assert(NSB * 16 <= T, "the middle section is one round");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lines = reinterpret_cast<float2*>(smem);
    float2* tw0 = lines + C * R0 * kWrSB;
    float2* twl = tw0 + (R0 - 1) * kWrS;
    uint8_t* stage = reinterpret_cast<uint8_t*>(twl + 256);
    const int rowbytes = cols * CH;
    const int stage_row = (rowbytes + 15) / 16 * 16;

    const int tid = threadIdx.x;
    for (int i = tid; i < (R0 - 1) * kWrS; i += T) tw0[i] = tw0g[i];
    wr_twl_fill(twl, w256, tid, T);
    const int lane16 = (tid >> 2) & 15, sbi = (tid >> 6) * 4 + (tid & 3);   // wr_transpose16's lane numbering
    const bool mid_on = sbi < NSB;
#ifndef WR_ROW_TWL
#define WR_ROW_TWL 0      // 1: the 256-point twiddles of the middle section are read from LDS where used; 0: 30 registers (measured: 0.8 us per 4K frame faster)
#endif
    WrMid<(WR_ROW_TWL != 0)> wm;
    wr_mid_load<R0>(wm, lane16, mid_on ? sbi % R0 : 0, w256, mult, twl);

    const int c = __builtin_amdgcn_readfirstlane(tid >> 8);     // the channel is the same for a whole wave: scalar
    const int j = tid & 255;
    const size_t plane = static_cast<size_t>(tile.plane_strips > 0 ? tile.plane_strips : nstrips) * npairs * (2 * G);
    const int strip_step = npairs * (2 * G);                     // floats between the same pair of neighbouring strips
    // per-thread constants (loop invariant, a handful of registers): byte offset of column j - pad inside a (channel, pair)
    // record set -- interior rounds add k * 32 strips -- and of the reflected columns of the edge rounds (~0u: zero)
    const int jm = j - pad;
    const unsigned off_in = static_cast<unsigned>(((jm >> LG) * strip_step + 2 * (jm & (G - 1))) * 4);
    unsigned off_e[NE];
#pragma unroll
    for (int e = 0; e < ELO + EHI; ++e) {
        const int k = e < ELO ? e : R0 - EHI + (e - ELO);
        const int x = wr_reflect(j + k * kWrS, pad, cols);
        off_e[e] = x < 0 ? ~0u : static_cast<unsigned>(((x >> LG) * strip_step + 2 * (x & (G - 1))) * 4);
    }
    const unsigned round_step = static_cast<unsigned>((kWrS / G) * strip_step * 4);   // bytes between rounds: 256 columns = 32 (64) strips
    const int st_off = jm * CH + c;                              // stage byte of column j - pad, row slot 0; slot 1 at + stage_row

    typedef float wr_f32x2 __attribute__((ext_vector_type(2)));
    wr_f32x2 pf[R0];
    const int upf = tile.tn > 0 ? tile.tn : npairs;             // units (row pairs) per frame
    auto unit_base = [&](int uu) -> const char* {
        const int ff = uu / upf, tt = uu - ff * upf + tile.t0;
        return reinterpret_cast<const char*>(inter + (static_cast<size_t>(ff) * CH + c) * plane + static_cast<size_t>(tt) * (2 * G));
    };
    // rounds [k0, k1) of the unit whose (channel, pair) records start at `base`
    auto issue_part = [&](const char* base, int k0, int k1) {
        const unsigned off_in_ = wr_opaque(off_in);
#pragma unroll
        for (int k = 0; k < R0; ++k) {
            if (k < k0 || k >= k1) continue;
            unsigned off;
            if (k >= ELO && k < R0 - EHI) off = off_in_ + k * round_step;
            else { off = off_e[k < ELO ? k : ELO + (k - (R0 - EHI))]; off = off == ~0u ? 0u : off; }      // unconditional loads, masked at use
#ifdef WR_ABL_NOLOAD      // ablation build (timing only, results are wrong): no global reads
            pf[k] = wr_f32x2{ static_cast<float>(off & 255u), 1.f };
#else
            pf[k] = *reinterpret_cast<const wr_f32x2*>(base + off);
#endif
        }
    };
    auto issue_unit = [&](int uu) { issue_part(unit_base(uu), 0, R0); };
    constexpr int QP = (R0 + 3) / 4;      // rounds per quarter
    auto claim_unit = [&]() {
#pragma unroll
        for (int k = 0; k < R0; ++k) asm volatile("" ::"v"(pf[k].x), "v"(pf[k].y));
    };
    // tiled images: the row pass's Nyquist-slot term (-1)^(x + pad) h(r) of the unit's two rows rides in the rounding constant
    const int ypar = tile.ypar < 0 ? (pad & 1) : tile.ypar;
    const int rows_out = tile.rows_full > 0 ? tile.rows_full : rows;
    const float sxf = ((jm + tile.xpar) & 1) ? -1.f : 1.f;       // (256 k is even: one sign per thread)
    auto half_of = [&](int uu, int slot) -> float {
        if (tile.h == nullptr) return 0.5f;                      // uniform
        const int ff = uu / upf, tt = uu - ff * upf + tile.t0;
        int r = tile.row_base + ff * tile.band_step + 2 * tt - ypar + slot;
        r = r < 0 ? 0 : (r >= rows_out ? rows_out - 1 : r);
        return 0.5f + sxf * tile.h[(static_cast<size_t>(tile.band_step > 0 ? 0 : ff) * CH + c) * rows_out + r];
    };
    // inverse pass 0 of the unit in `lines` -> "+0.5f, truncate" -> byte stage
    auto inverse_pass0 = [&](int uu) {
        float2 v[R0];
        const int j_ = wr_opaque(j);
        const float ha = half_of(uu, 0), hb = half_of(uu, 1);
        wr_ip0_load<R0>(v, lines + c * (R0 * kWrSB), tw0, j_);
        uint8_t* const st_a_ = stage + wr_opaque(st_off);
#pragma unroll
        for (int k = 0; k < R0; ++k) {
            bool ok = true;
            if (k < ELO || k >= R0 - EHI) { const int x = jm + k * kWrS; ok = x >= 0 && x < cols; }
            if (ok) {
                uint8_t* s = st_a_ + k * (kWrS * CH);
                s[0] = static_cast<uint8_t>(static_cast<int>(v[k].x + ha));
                s[stage_row] = static_cast<uint8_t>(static_cast<int>(v[k].y + hb));
            }
        }
    };
    // the two image rows of unit (f, t) out of the stage
    auto write_out = [&](int f, int t) {
        if (tile.vr1 > 0 || tile.vc1 > 0 || tile.dst_pitch > 0) {
            // a band / tile of a larger image: image rows [vr0, vr1) and the call's columns [vc0, vc1) only, rows dst_pitch apart;
            // 16-byte pieces from the (aligned) stage to wherever they belong, then the odd bytes
            const int bofs = f * tile.band_step;                  // this band's offset in image rows (0: frames are images)
            int vr1 = (tile.vr1 > 0 ? tile.vr1 : tile.row_base + rows) + bofs;
            if (tile.band_step > 0 && tile.vr_max > 0 && vr1 > tile.vr_max) vr1 = tile.vr_max;
            const int vr0 = tile.vr0 + bofs, vc1 = tile.vc1 > 0 ? tile.vc1 : cols;
            const size_t pitch = tile.dst_pitch > 0 ? static_cast<size_t>(tile.dst_pitch) : static_cast<size_t>(rowbytes);
            const int g0 = tile.row_base + bofs + 2 * t - ypar;  // image row of slot 0
            const int fimg = tile.band_step > 0 ? 0 : f;
            const int nb = (vc1 - tile.vc0) * CH, n16 = nb >> 4;
            for (int rb = 0; rb < 2; ++rb) {
                const int gr = g0 + rb;
                if (gr < vr0 || gr >= vr1) continue;            // uniform
                uint8_t* const o = dst + (static_cast<size_t>(fimg) * rows_out + gr) * pitch + static_cast<size_t>(tile.dst_x0 + tile.vc0) * CH;
                const uint8_t* const sp = stage + rb * stage_row + tile.vc0 * CH;
                typedef unsigned int wr_u32x4 __attribute__((ext_vector_type(4), aligned(1)));
                for (int i = tid; i < n16; i += T) {
                    const uint4 q = *reinterpret_cast<const uint4*>(sp + 16 * i);
                    const wr_u32x4 w = { q.x, q.y, q.z, q.w };
                    *reinterpret_cast<wr_u32x4*>(o + 16 * i) = w;
                }
                for (int i = 16 * n16 + tid; i < nb; i += T) o[i] = sp[i];
            }
            return;
        }
        const int r0 = 2 * t - (pad & 1);                        // slot 0 row (may be -1), slot 1 row r0 + 1 (may be rows)
        uint8_t* const out0 = dst + (static_cast<size_t>(f) * rows + (r0 < 0 ? 0 : r0)) * rowbytes;
        const bool ok0 = r0 >= 0, ok1 = r0 + 1 < rows;
        if (aligned16) {
            const int nchunk = rowbytes >> 4;
            for (int idx = tid; idx < 2 * nchunk; idx += T) {
                const int rb = idx >= nchunk ? 1 : 0, i = idx - rb * nchunk;
#ifdef WR_ABL_NOSTORE     // ablation build: no global writes, values kept alive
                const uint4 q_ = *reinterpret_cast<const uint4*>(stage + rb * stage_row + 16 * i);
                asm volatile("" ::"v"(q_.x), "v"(q_.y), "v"(q_.z), "v"(q_.w), "v"(out0));
                if (false)
#else
                if (rb ? ok1 : ok0)
#endif
                    *reinterpret_cast<uint4*>(out0 + static_cast<size_t>(rb && ok0 ? rowbytes : 0) + 16 * i) =
                        *reinterpret_cast<const uint4*>(stage + rb * stage_row + 16 * i);
            }
        } else {
            for (int idx = tid; idx < 2 * rowbytes; idx += T) {
                const int rb = idx >= rowbytes ? 1 : 0, i = idx - rb * rowbytes;
                if (rb ? ok1 : ok0) out0[static_cast<size_t>(rb && ok0 ? rowbytes : 0) + i] = stage[rb * stage_row + i];
            }
        }
    };

    WR_STAMP_DECL;
    const WrWalk walk = wr_walk(nunits);
    if (walk.begin < walk.end) issue_unit(walk.begin);
    claim_unit();
    __syncthreads();
    WR_STAMP(7);      // prologue
    int pu = -1;      // previous unit (its inverse pass 0 and write-out are still due)
    for (int u = walk.begin; u < walk.end; u += walk.step) {
        // ---- phase A: inverse pass 0 of the previous unit, pass 0 of this one (in place, same elements per thread)
        if (pu >= 0) inverse_pass0(pu);
        __builtin_amdgcn_sched_barrier(0);      // one half after the other: together they do not fit the register budget
        WR_STAMP(4);      // inverse pass 0
        {
            float2 v[R0];
#pragma unroll
            for (int k = 0; k < R0; ++k) {
                bool ok = true;
                if (k < ELO || k >= R0 - EHI) ok = off_e[k < ELO ? k : ELO + (k - (R0 - EHI))] != ~0u;
                v[k] = ok ? make_float2(pf[k].x, pf[k].y) : make_float2(0.f, 0.f);
            }
            wr_p0_store<R0>(v, lines + c * (R0 * kWrSB), tw0, wr_opaque(j));
        }
        WR_STAMP(0);      // pass 0
        __syncthreads();
        WR_STAMP(1);
        // ---- phase B: write-out of the previous unit (the stage is complete), middle of this one
#ifndef WR_ROW_STORE_LATE
        if (pu >= 0) write_out(pu / upf, pu % upf + tile.t0);
#endif
        WR_STAMP(6);      // write-out
        // the next unit's input is requested a quarter at a time from inside the middle section
        const int un = u + walk.step;
        const char* const nbase = unit_base(un < walk.end ? un : u);          // past the end: this unit again (the data is not used)
        auto hook = wr_pinned([&](int q) { issue_part(nbase, q * QP, (q + 1) * QP < R0 ? (q + 1) * QP : R0); });
        if (WR_MID_ON(mid_on)) wr_middle(lines + wr_opaque(sbi) * kWrSB, wm, wr_opaque(lane16), hook);
        else { hook(0, 0.f); hook(1, 0.f); hook(2, 0.f); hook(3, 0.f); }
#ifdef WR_ROW_STORE_LATE
        if (pu >= 0) write_out(pu / upf, pu % upf + tile.t0);      // behind the requests for the next unit's rows
#endif
        WR_STAMP(2);      // middle
        // (no wait for the requested rows here: the barrier must not wait on memory latency.  Nothing else is issued to
        //  memory before pass 0 consumes them, so the wait the compiler places there counts exactly these loads.)
        __syncthreads();
        WR_STAMP(3);
        pu = u;
    }
    if (pu >= 0) {
        inverse_pass0(pu);
        __syncthreads();
        write_out(pu / upf, pu % upf + tile.t0);
    }
    WR_STAMP_FLUSH(mult, R0 * kWrS, T / 64);
}

// ======================================================================================
// complex lines in, convolved complex lines out (tests, and the batched line API)
// ======================================================================================
template <int R0, int C, int T>
__global__ __launch_bounds__(T) void wr_lines_kernel(const float2* __restrict__ in, float2* __restrict__ out, int nlines,
                                                     const float2* __restrict__ w256, const float2* __restrict__ tw0g, const float* __restrict__ mult)
{
    constexpr int NSB = C * R0, N = R0 * kWrS;
    constexpr int total0 = C * kWrS, IT0 = (total0 + T - 1) / T;
    static_assert(NSB * 16 <= T && T % 64 == 0, "the middle section is one round");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lines = reinterpret_cast<float2*>(smem);
    float2* tw0 = lines + C * R0 * kWrSB;
    const int tid = threadIdx.x;
    for (int i = tid; i < (R0 - 1) * kWrS; i += T) tw0[i] = tw0g[i];
    const int lane16 = (tid >> 2) & 15, sbi = (tid >> 6) * 4 + (tid & 3);   // wr_transpose16's lane numbering
    const bool mid_on = sbi < NSB;
    WrMid<false> wm;
    wr_mid_load<R0>(wm, lane16, mid_on ? sbi % R0 : 0, w256, mult);
    __syncthreads();
    for (int l0 = blockIdx.x * C; l0 < nlines; l0 += gridDim.x * C) {
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int g = tid + T * it;
            if (g < total0) {
                const int l = g >> 8, j = g & 255;
                float2 v[R0];
                const int line = l0 + l < nlines ? l0 + l : nlines - 1;
#pragma unroll
                for (int k = 0; k < R0; ++k) v[k] = in[static_cast<size_t>(line) * N + j + k * kWrS];
                wr_p0_store<R0>(v, lines + l * (R0 * kWrSB), tw0, j);
            }
        }
        __syncthreads();
        if (WR_MID_ON(mid_on)) wr_middle(lines + wr_opaque(sbi) * kWrSB, wm, wr_opaque(lane16));
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int g = tid + T * it;
            if (g < total0) {
                const int l = g >> 8, j = g & 255;
                float2 v[R0];
                wr_ip0_load<R0>(v, lines + l * (R0 * kWrSB), tw0, j);
                if (l0 + l < nlines) {
#pragma unroll
                    for (int k = 0; k < R0; ++k) out[static_cast<size_t>(l0 + l) * N + j + k * kWrS] = v[k];
                }
            }
        }
        __syncthreads();
    }
}

// ---- launchers ---------------------------------------------------------------------------
struct WrEntry {
    int r0;          // N = 256 * r0
    int threads;
    int g;           // column role: columns per strip of the intermediate it writes (8, or 4 for the C = 2 kernels); row role: 0
    // columns first: u8 frames -> intermediate
    hipError_t (*col_u8)(hipStream_t, const uint8_t* src, float* inter, int rows, int cols, int pad, int nframes, int num_cus,
                         const float2* w256, const float2* tw0, const float* mult, WrColTerm term);
    size_t (*col_lds)(int rows);
    // rows second: intermediate -> u8 frames
    hipError_t (*row_u8)(hipStream_t, const float* inter, uint8_t* dst, int rows, int cols, int pad, int nframes, int num_cus,
                         const float2* w256, const float2* tw0, const float* mult, WrRowTile tile);
    size_t (*row_lds)(int cols);
    // complex lines of length N (in == out allowed)
    hipError_t (*lines)(hipStream_t, const float2* in, float2* out, int nlines, int num_cus, const float2* w256, const float2* tw0, const float* mult);
};

inline int wr_npairs(int rows, int pad) { return ((rows - 1 + pad) >> 1) - (pad >> 1) + 1; }
inline size_t wr_frame_floats(int rows, int cols, int pad, int g = 8) { return static_cast<size_t>((cols + g - 1) / g) * wr_npairs(rows, pad) * (2 * g) * 3; }

inline int wr_balanced_grid(int units, int slots)
{
    if (units <= slots) return units;
    const int rounds = (units + slots - 1) / slots;
    return (units + rounds - 1) / rounds;
}

template <class K> hipError_t wr_set_lds(K kern, size_t lds)
{
    if (lds > 64 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    return hipSuccess;
}

template <int R0, int C, int T>
hipError_t wr_launch_col_u8(hipStream_t st, const uint8_t* src, float* inter, int rows, int cols, int pad, int nframes, int num_cus,
                            const float2* w256, const float2* tw0, const float* mult, WrColTerm term)
{
    if (rows + 2 * pad > R0 * kWrS || pad > rows - 1) return hipErrorInvalidValue;
    const size_t lds = wr_col_lds<R0, C>(rows);
    // rounds k with 256 k >= pad and 256 (k + 1) <= pad + rows are interior
    const int elo = (pad + kWrS - 1) / kWrS, ehi = R0 - (pad + rows) / kWrS;
    auto kern = (elo <= 1 && ehi <= 1) ? wr_colpass_u8<R0, C, T, 1, 1> : wr_colpass_u8<R0, C, T, R0, 0>;
    if (hipError_t e = wr_set_lds(kern, lds); e != hipSuccess) return e;
    const int nstrips = (cols + 2 * C - 1) / (2 * C), npairs = wr_npairs(rows, pad), nunits = nstrips * nframes;
    // 32-bit float offsets inside a channel plane
    if (static_cast<size_t>(nstrips) * npairs * 4 * C >= (static_cast<size_t>(1) << 30)) return hipErrorInvalidValue;
    // (the strip's 8-byte pieces may sit at any byte address: gfx950 takes unaligned vector loads; only a ragged last strip goes byte by byte)
    const int aligned8 = 1;
    const int grid = wr_balanced_grid(nunits, num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, st, src, inter, rows, cols, pad, npairs, nstrips, nunits, aligned8, w256, tw0, mult, term);
    return hipGetLastError();
}

template <int R0, int T>
hipError_t wr_launch_row_u8(hipStream_t st, const float* inter, uint8_t* dst, int rows, int cols, int pad, int nframes, int num_cus,
                            const float2* w256, const float2* tw0, const float* mult, WrRowTile tile)
{
    if (cols + 2 * pad > R0 * kWrS || pad > cols - 1) return hipErrorInvalidValue;
    const size_t lds = wr_row_lds<R0>(cols);
    const int elo = (pad + kWrS - 1) / kWrS, ehi = R0 - (pad + cols) / kWrS;
    auto kern = tile.g == 4 ? ((elo <= 1 && ehi <= 1) ? wr_rowpass_u8<R0, T, 1, 1, 4> : wr_rowpass_u8<R0, T, R0, 0, 4>)
                            : ((elo <= 1 && ehi <= 1) ? wr_rowpass_u8<R0, T, 1, 1, 8> : wr_rowpass_u8<R0, T, R0, 0, 8>);
    if (tile.g != 4 && tile.g != 8) return hipErrorInvalidValue;
    if (hipError_t e = wr_set_lds(kern, lds); e != hipSuccess) return e;
    // (a band of a tiled image: the rows pair up by the COLUMN call's padding, tile.ypar)
    const int nstrips = (cols + tile.g - 1) / tile.g, npairs = tile.ypar < 0 ? wr_npairs(rows, pad) : ((rows - 1 + tile.ypar) >> 1) + 1;
    if (tile.tn > 0 && (tile.t0 < 0 || tile.t0 + tile.tn > npairs)) return hipErrorInvalidValue;
    const int nunits = (tile.tn > 0 ? tile.tn : npairs) * nframes;
    if (static_cast<size_t>(tile.plane_strips > 0 ? tile.plane_strips : nstrips) * npairs * (2 * tile.g) >= (static_cast<size_t>(1) << 30)) return hipErrorInvalidValue;
    const int aligned16 = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && ((static_cast<size_t>(cols) * 3) & 15) == 0) ? 1 : 0;
    const int grid = wr_balanced_grid(nunits, num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, st, inter, dst, rows, cols, pad, npairs, nstrips, nunits, aligned16, w256, tw0, mult, tile);
    return hipGetLastError();
}

template <int R0, int C, int T>
hipError_t wr_launch_lines(hipStream_t st, const float2* in, float2* out, int nlines, int num_cus, const float2* w256, const float2* tw0, const float* mult)
{
    const size_t lds = wr_lines_bytes<R0, C>() + wr_tw0_bytes<R0>();
    auto kern = wr_lines_kernel<R0, C, T>;
    if (hipError_t e = wr_set_lds(kern, lds); e != hipSuccess) return e;
    const int units = (nlines + C - 1) / C;
    if (units <= 0) return hipSuccess;
    const int grid = units < num_cus ? units : num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, st, in, out, nlines, w256, tw0, mult);
    return hipGetLastError();
}


}  // namespace blur_amd

// one translation unit per (R0, role):  BLUR_WR_COL(9, 4, 576)   BLUR_WR_ROW(16, 768)
#define BLUR_WR_COL(R0_, C_, T_)                                                                  \
    namespace blur_amd {                                                                          \
    const WrEntry* wr_col_entry_##R0_()                                                           \
    {                                                                                             \
        static const WrEntry e = { R0_, T_, 2 * C_, wr_launch_col_u8<R0_, C_, T_>, wr_col_lds<R0_, C_>, nullptr, nullptr, \
                                   wr_launch_lines<R0_, C_, T_> };                                \
        return &e;                                                                                \
    }                                                                                             \
    }
#define BLUR_WR_ROW(R0_, T_)                                                                      \
    namespace blur_amd {                                                                          \
    const WrEntry* wr_row_entry_##R0_()                                                           \
    {                                                                                             \
        static const WrEntry e = { R0_, T_, 0, nullptr, nullptr, wr_launch_row_u8<R0_, T_>, wr_row_lds<R0_>, \
                                   wr_launch_lines<R0_, 3, T_> };                                 \
        return &e;                                                                                \
    }                                                                                             \
    }
