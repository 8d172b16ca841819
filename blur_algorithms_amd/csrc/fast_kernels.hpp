// fast_kernels.hpp -- compile-time specialised row / column kernels for the FFT lengths the
// BASELINE configs land on.  Same algorithm and data path as the generic kernels in
// engine.hip (see there), but the plan (N and its radix sequence) is a template argument so
// that every stride, trip count and table offset is a constant.  Both kernels follow one recipe:
//   * ONE persistent workgroup of 576-768 threads per CU works on several complex lines at once (row pass: the three
//     channel lines of a row pair; column pass: the four lines of a strip of 8 columns), every pass flattened over
//     (line, butterfly), the radix order chosen so that a pass is one round of butterflies wherever possible;
//   * the next unit's input is requested into registers while this unit is transformed, claimed before this unit's
//     stores are issued (vmcnt retires in order), and committed to LDS when the line buffer is free;
//   * pass 0's twiddles live in registers or (plan flag 4) in LDS, the small twiddle tables of the inner passes and
//     the pointwise multipliers of the fused middle pass in LDS;
//   * row pass: u8 rows in (deinterleave + reflect-101 pad fused, Utils.hpp:159-184 / Source.cpp:525-529), cropped
//     float rows out (Source.cpp:536); column pass: the last inverse pass applies interleave_BGR's "+0.5f, truncate"
//     (Utils.hpp:189,204-206) into an LDS pixel stage that leaves as whole pixels.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_engine.hpp"

// tuning knobs (overridable per build for A/B runs: tools/build_variant.sh with VFLAGS)
#ifndef FK_INNER_BATCH
#define FK_INNER_BATCH 2            // butterflies of one thread whose LDS reads are issued together (inner passes)
#endif
#ifndef FK_BATCH_MAX_R
#define FK_BATCH_MAX_R 5            // largest radix that is batched (radix 10 in pairs spills at the 168-VGPR budget)
#endif
#ifndef FK_INNER_UNROLL
#define FK_INNER_UNROLL 1           // butterflies of an inner / middle pass a thread keeps in flight together
#endif
#ifndef FK_HOIST_MAX_R
#define FK_HOIST_MAX_R 16           // inner passes up to this radix read their twiddles up front
#endif
#ifndef FK_COL_PREFETCH
#define FK_COL_PREFETCH 1           // column kernel, strip layout: load the next task's strip into registers during the passes
#endif
#ifndef FK_GATHER_UNROLL
#define FK_GATHER_UNROLL 4          // independent strip-gather loads a thread keeps in flight (column kernel)
#endif
// ablation builds of the in-LDS passes (timing only, results are wrong): -DFK_ABL_NOLDSW drops the LDS stores of the
// inner and middle passes (two adds per element keep the arithmetic alive), -DFK_ABL_NOMATH drops the butterflies
#ifdef FK_ABL_NOLDSW
#define FK_ST(dst, val) do { const float2 v_ = (val); fk_abl_acc += v_.x + v_.y; } while (0)
#define FK_ABL_DECL float fk_abl_acc = 0.f
#define FK_ABL_SINK(ptr) do { if (fk_abl_acc == 1.2345e30f) (ptr)[0] = make_float2(fk_abl_acc, 0.f); } while (0)
#else
#define FK_ST(dst, val) (dst) = (val)
#define FK_ABL_DECL do { } while (0)
#define FK_ABL_SINK(ptr) do { } while (0)
#endif
#ifdef FK_ABL_NOMATH
#define FK_BFLY(R, INV, v) do { } while (0)
#else
#define FK_BFLY(R, INV, v) Bfly<R, INV>::run(v)
#endif
#define FK_PRAGMA(x) _Pragma(#x)
#define FK_UNROLL(n) FK_PRAGMA(unroll n)

namespace blur_amd {

// Diagnostic builds only (-DFK_STAMPS): shader-clock stamps around the phases of a line, summed
// per workgroup and written behind the multiplier table (the host allocates a tail for it).
// No stamp executes in a normal build.
constexpr int kStampSlots = 8;
constexpr int kStampTailFloats = 1 << 16;
#ifdef FK_STAMPS
#define FK_STAMP(i)                                                   \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                           \
        __builtin_amdgcn_sched_barrier(0);                            \
        st_acc[i] += t_ - st_prev;                                    \
        st_prev = t_;                                                 \
    } while (0)
#else
#define FK_STAMP(i) do { } while (0)
#endif

// ---- static plan ---------------------------------------------------------------------
// FLAGS_ (bits):
//   1  LDS line padding: one spare element per 32 (element i at i + (i >> 5)) instead of none.  Only plans whose inner
//      passes stride by a multiple of 32 elements need it, and it costs three VALU instructions per LDS address.
//   2  register diet: inner passes of radix > 10 read their twiddles where they are used instead of up front
//   4  column kernel: pass 0's twiddles live in LDS (behind the pixel stage) instead of registers
//  32  row pass by fast_rowpass3_u8: the three channels of a row pair are transformed together by one workgroup per
//      CU (pass-0 twiddles in LDS), the way the column kernel treats the four lines of a strip
//   8  column kernel: compile for 3 waves per SIMD (<= 168 VGPRs) although one workgroup alone would not need it, so
//      that TWO workgroups fit a CU when their LDS does (short lines)
template <int N_, int FLAGS_, int... Rs> struct StaticPlan {
    static constexpr int N = N_;
    static constexpr int PAD = FLAGS_ & 1;
    static constexpr int hoist_max = (FLAGS_ & 2) ? (FK_HOIST_MAX_R < 10 ? FK_HOIST_MAX_R : 10) : FK_HOIST_MAX_R;
    static constexpr bool tw0_lds = (FLAGS_ & 4) != 0;
    static constexpr int col_min_waves = (FLAGS_ & 8) ? 3 : 1;
#ifndef FK_COL_OPAQUE
#define FK_COL_OPAQUE 15
#endif
    static constexpr int col_opaque = (FLAGS_ & 64) ? FK_COL_OPAQUE : 0;   // 64: keep per-thread offsets out of the loop-invariant set (bit per site)
    static __host__ __device__ constexpr int at(int i) { return PAD ? i + (i >> 5) : i; }
    static __host__ __device__ constexpr int zs() { return at(N_) + 1; }
    static constexpr int P = sizeof...(Rs);
    static constexpr int R[P] = { Rs... };
    static constexpr int m(int i) { int len = N; for (int k = 0; k <= i; ++k) len /= R[k]; return len; }
    static constexpr int nb(int i) { return N / R[i]; }
    // offset of pass i's twiddles inside the plan's table (host layout, make_plan_radices)
    static constexpr int tw_off(int i) { int off = 0; for (int k = 0; k < i; ++k) if (m(k) > 1) off += (R[k] - 1) * m(k); return off; }
    // inner passes 1..P-2 go to LDS
    static constexpr int lds_tw_begin() { return tw_off(1); }
    static constexpr int lds_tw_count() { return P > 2 ? tw_off(P - 1) - tw_off(1) : 0; }
    static constexpr bool valid() { int p = 1; for (int k = 0; k < P; ++k) p *= R[k]; return p == N && P >= 2 && P <= kMaxPassesDev; }
};

struct FastEntry {
    int n;
    int npass;
    int radix[kMaxPassesDev];
    // nframes frames back to back (u8: rows*cols*3 bytes each).
    // Layout of the float intermediate, per frame and channel:
    //   tile_w == 0: one row-major plane [rows][cols]
    //   tile_w == 8: strips of 8 columns, each strip contiguous and row-PAIR interleaved:
    //     [strip][row pair q][column 0..7][row 2q, row 2q+1]
    //     (the row kernel's complex value (row a, row b) of one column is ONE 8-byte store, eight
    //      neighbouring columns make a whole 64-byte sector; the column kernel streams the strip
    //      with fully coalesced 16-byte loads: two columns x two rows each)
    hipError_t (*row_u8)(hipStream_t, const uint8_t* src, float* planes, int rows, int cols, int pad, int nframes, int tile_w,
                         const float2* tw, const float* mperm);
    // C = complex lines per workgroup (strip width / 2): 2 or 4; tiled: planes are in the strip layout with tile_w = 2C
    hipError_t (*col_u8)(hipStream_t, const float* planes, uint8_t* dst, int rows, int cols, int pad, int nframes, int tiled,
                         const float2* tw, const float* mperm, int C);
    size_t (*col_lds_bytes)(int rows, int C);
};

__device__ __forceinline__ int fk_reflect_src(int p, int pad, int len)
{
    const int i = p - pad;
    if (i < 0) return -i;
    if (i < len) return i;
    if (i < len + pad) return 2 * (len - 1) - i;
    return -1;
}

// ---- inner passes on LDS, flattened over (line, butterfly) ----------------------------------
// FK_INNER_BATCH butterflies of a thread are read together (data and twiddles of all of them
// before the first use), so that one LDS latency is paid per batch and the second butterfly's
// reads are in flight while the first is computed.  Radices above PL::hoist_max keep batch 1 and
// read their twiddles where they are used (register pressure).
template <class PL, int I, int C, int T, bool INV>
__device__ __forceinline__ void fk_inner_pass(float2* z, int zs, const float2* twl)
{
    FK_ABL_DECL;
    constexpr int R = PL::R[I], m = PL::m(I), nb = PL::nb(I), total = nb * C;
    constexpr int off = PL::tw_off(I) - PL::lds_tw_begin();
    constexpr bool hoist = R <= PL::hoist_max;
    constexpr int B = (hoist && R <= FK_BATCH_MAX_R && total > T) ? FK_INNER_BATCH : 1;
#pragma unroll 1
    for (int g0 = threadIdx.x; g0 < total; g0 += T * B) {
        float2 v[B][R], w[B][R];
        int base[B], jj[B];
        bool act[B];
        float2* zc[B];
#pragma unroll
        for (int bi = 0; bi < B; ++bi) {
            const int g = g0 + bi * T;
            act[bi] = g < total;
            const int gg = act[bi] ? g : total - 1;
            const int c = gg / nb, b = gg - c * nb;
            const int blk = b / m;
            jj[bi] = b - blk * m;
            base[bi] = blk * (R * m) + jj[bi];
            zc[bi] = z + c * zs;
#pragma unroll
            for (int k = 0; k < R; ++k) v[bi][k] = zc[bi][PL::at(base[bi] + k * m)];
            if constexpr (hoist) {
#pragma unroll
                for (int q = 1; q < R; ++q) w[bi][q] = twl[off + (q - 1) * m + jj[bi]];
            }
        }
        if constexpr (hoist) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bi = 0; bi < B; ++bi) {
            if constexpr (!INV) {
                FK_BFLY(R, false, v[bi]);
                if (act[bi]) {
                    FK_ST(zc[bi][PL::at(base[bi])], v[bi][0]);
#pragma unroll
                    for (int q = 1; q < R; ++q)
                        FK_ST(zc[bi][PL::at(base[bi] + q * m)], cmul(v[bi][q], hoist ? w[bi][q] : twl[off + (q - 1) * m + jj[bi]]));
                }
            } else {
#pragma unroll
                for (int q = 1; q < R; ++q) v[bi][q] = cmulc(v[bi][q], hoist ? w[bi][q] : twl[off + (q - 1) * m + jj[bi]]);
                FK_BFLY(R, true, v[bi]);
                if (act[bi]) {
#pragma unroll
                    for (int k = 0; k < R; ++k) FK_ST(zc[bi][PL::at(base[bi] + k * m)], v[bi][k]);
                }
            }
        }
    }
    FK_ABL_SINK(z);
}

template <class PL, int I, int C, int T, bool INV>
__device__ __forceinline__ void fk_inner_passes(float2* z, int zs, const float2* twl)
{
    // forward: I = 1 .. P-2 ascending; inverse: P-2 .. 1 descending
    if constexpr (I >= 1 && I <= PL::P - 2) {
        fk_inner_pass<PL, I, C, T, INV>(z, zs, twl);
        __syncthreads();
        fk_inner_passes<PL, (INV ? I - 1 : I + 1), C, T, INV>(z, zs, twl);
    }
}

// fused middle: last forward pass (m == 1) * multipliers * first inverse pass,
// flattened over (line, butterfly), multipliers read from LDS
template <class PL, int T, int C>
__device__ __forceinline__ void fk_mid_lds(float2* z, int zs, const float* __restrict__ mpl)
{
    FK_ABL_DECL;
    constexpr int R = PL::R[PL::P - 1], nb = PL::nb(PL::P - 1), total = nb * C;
    constexpr int B = (R <= FK_BATCH_MAX_R && total > T) ? FK_INNER_BATCH : 1;
#pragma unroll 1
    for (int g0 = threadIdx.x; g0 < total; g0 += T * B) {
        float2 v[B][R];
        float mm[B][R];
        float2* zc[B];
        int base[B];
        bool act[B];
#pragma unroll
        for (int bi = 0; bi < B; ++bi) {
            const int g = g0 + bi * T;
            act[bi] = g < total;
            const int gg = act[bi] ? g : total - 1;
            const int c = gg / nb, b = gg - c * nb;
            zc[bi] = z + c * zs;
            base[bi] = b * R;
#pragma unroll
            for (int k = 0; k < R; ++k) v[bi][k] = zc[bi][PL::at(base[bi] + k)];
#pragma unroll
            for (int q = 0; q < R; ++q) mm[bi][q] = mpl[base[bi] + q];
        }
        if constexpr (R <= PL::hoist_max) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bi = 0; bi < B; ++bi) {
            FK_BFLY(R, false, v[bi]);
#pragma unroll
            for (int q = 0; q < R; ++q) v[bi][q] = cscale(v[bi][q], mm[bi][q]);
            FK_BFLY(R, true, v[bi]);
            if (act[bi]) {
#pragma unroll
                for (int k = 0; k < R; ++k) FK_ST(zc[bi][PL::at(base[bi] + k)], v[bi][k]);
            }
        }
    }
    FK_ABL_SINK(z);
}

// pass-0 twiddles in registers: butterfly j = tid (+ T*it)
template <class PL, int T> struct Pass0Regs {
    static constexpr int R = PL::R[0];
    static constexpr int m = PL::m(0);          // == nb(0): one block of length N
    static constexpr int IT = (m + T - 1) / T;
    float2 w[IT][R];                            // w[.][0] unused
    __device__ __forceinline__ void load(const float2* __restrict__ tw, int j0)
    {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int j = j0 + T * it;
#pragma unroll
            for (int q = 1; q < R; ++q) w[it][q] = tw[(q - 1) * m + (j < m ? j : m - 1)];   // unconditional: loads must not serialise
        }
    }
};

// ======================================================================================
// row pass
// ======================================================================================
// One workgroup per CU transforms the CH complex lines of a row pair (one per channel) at once, flattened over
// (channel, butterfly) in every pass -- the column kernel's recipe: many waves in ONE workgroup, every pass close to
// one full round of butterflies, pass 0's twiddles in LDS so that no pass pins registers across the others.
// LDS: CH lines | inner twiddles | pass-0 twiddles (plan flag 4 only; else registers) | multipliers.
template <class PL, int CH> __host__ __device__ constexpr size_t fk_row3_lds()
{
    return (static_cast<size_t>(CH) * PL::zs() + ((PL::lds_tw_count() + 1) & ~1) + (PL::tw0_lds ? static_cast<size_t>(PL::R[0] - 1) * PL::m(0) : 0)) * sizeof(float2) +
           static_cast<size_t>(PL::N) * sizeof(float);
}

template <class PL, int T, int CH, int tile_shift, bool STAGED>
__global__ __launch_bounds__(T) void fast_rowpass3_u8(const uint8_t* __restrict__ src, float* __restrict__ planes,
                                                      int rows, int cols, int pad, int npairs, int nunits,
                                                      const float2* __restrict__ tw, const float* __restrict__ mperm)
{
    static_assert(PL::valid(), "radices do not multiply to N");
    constexpr int N = PL::N, P = PL::P;
    constexpr int R0 = PL::R[0], m0 = PL::m(0);
    constexpr int zs = PL::zs();
    constexpr int total0 = CH * m0, IT0 = (total0 + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* z = reinterpret_cast<float2*>(smem);
    float2* twl = z + CH * zs;
    float2* tw0l = twl + ((PL::lds_tw_count() + 1) & ~1);
    float* mpl = reinterpret_cast<float*>(tw0l + (PL::tw0_lds ? (R0 - 1) * m0 : 0));
    constexpr int tile_w = tile_shift ? 1 << tile_shift : 0;
    const size_t plane_elems = tile_shift ? static_cast<size_t>((cols + tile_w - 1) >> tile_shift) * npairs * (2 * tile_w)
                                          : static_cast<size_t>(rows) * cols;
    for (int i = threadIdx.x; i < PL::lds_tw_count(); i += T) twl[i] = tw[PL::lds_tw_begin() + i];
    if constexpr (PL::tw0_lds) {
        for (int i = threadIdx.x; i < (R0 - 1) * m0; i += T) tw0l[i] = tw[i];
    }
    for (int i = threadIdx.x; i < N; i += T) mpl[i] = mperm[i];
    // plan flag 4 clear: pass 0's twiddles in registers instead (butterfly (c, j) of pass 0 is this thread's in every unit)
    float2 w0[IT0][R0];
    if constexpr (!PL::tw0_lds) {
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            int g = threadIdx.x + T * it;
            g = g < total0 ? g : total0 - 1;
            const int j = g % m0;
#pragma unroll
            for (int q = 1; q < R0; ++q) w0[it][q] = tw[(q - 1) * m0 + j];
        }
    }
    auto p0w = [&](int it, int q, int j) -> float2 {
        if constexpr (PL::tw0_lds) return tw0l[(q - 1) * m0 + j];
        else return w0[it][q];
    };

    // STAGED (rows 16-byte aligned, checked by the launcher): the two u8 rows of a unit arrive by 16-byte loads
    // that were issued into registers while the previous unit was being transformed, are parked in the (then free)
    // line buffer, and every thread picks its 2*R0 bytes from there before the first butterfly overwrites them.
    const int rowbytes = cols * CH;
    uint8_t* const stage = reinterpret_cast<uint8_t*>(z);
    constexpr int KP = STAGED ? (2 * ((N * CH + 15) / 16) + T - 1) / T : 1;
    static_assert(!STAGED || 2 * static_cast<size_t>(N) * CH <= static_cast<size_t>(CH) * zs * sizeof(float2), "the stage fits the line buffer");
    typedef unsigned int fk_u32x4 __attribute__((ext_vector_type(4)));   // a plain vector value: stays in registers
    fk_u32x4 pfr[KP];
    auto issue_rows = [&](int uu) {
        const int ff = uu / npairs, pp = uu - ff * npairs;
        const uint8_t* a = src + (static_cast<size_t>(ff) * rows + 2 * pp) * rowbytes;
        const int nchunk = rowbytes >> 4;
        const int second = (2 * pp + 1 < rows) ? rowbytes : 0;      // a missing second row reads the first again (masked later)
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            int idx = threadIdx.x + T * k;
            idx = idx < 2 * nchunk ? idx : 2 * nchunk - 1;
            const int rb = idx >= nchunk ? 1 : 0;
#ifdef FK_ABL_NOLOAD    // ablation build: no global reads (timing only, results are wrong)
            pfr[k] = fk_u32x4{ static_cast<unsigned>(idx), static_cast<unsigned>(uu), 3u, static_cast<unsigned>(second) };
#else
            pfr[k] = *reinterpret_cast<const fk_u32x4*>(a + (rb ? second : 0) + 16 * (idx - rb * nchunk));
#endif
        }
    };
    // "claim": make the compiler wait for the prefetched rows HERE.  vmcnt retires in order and the number of
    // stores a unit issues is not known at compile time, so a wait for these loads that came after stores -- or at a
    // join of a path with pending loads and one without -- would drain every store in flight.  With the rows
    // claimed before the stores on every path, the loop carries only stores across its back edge.
    auto claim_rows = [&]() {
#pragma unroll
        for (int k = 0; k < KP; ++k) asm volatile("" ::"v"(pfr[k].x), "v"(pfr[k].y), "v"(pfr[k].z), "v"(pfr[k].w));
    };

    const int u_begin = static_cast<int>(static_cast<long long>(blockIdx.x) * nunits / gridDim.x);
    const int u_end = static_cast<int>(static_cast<long long>(blockIdx.x + 1) * nunits / gridDim.x);
    if constexpr (STAGED) {
        if (u_begin < u_end) issue_rows(u_begin);
        claim_rows();
    }
    for (int u = u_begin; u < u_end; ++u) {
        const int f = u / npairs, pair = u - f * npairs;
        const int r0 = 2 * pair;
        const bool two = r0 + 1 < rows;
        const uint8_t* row_a = src + (static_cast<size_t>(f) * rows + r0) * cols * CH;
        const uint8_t* row_b = row_a + (two ? static_cast<size_t>(cols) * CH : 0);
        float* out_u = planes + static_cast<size_t>(f) * plane_elems * CH +
                       (tile_shift ? static_cast<size_t>(pair) * (2 * tile_w) : static_cast<size_t>(r0) * cols);
        __syncthreads();   // the previous unit's readers are done with z (and the tables are visible)
        if constexpr (STAGED) {
            const int nchunk = rowbytes >> 4;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int idx = threadIdx.x + T * k;
                if (idx < 2 * nchunk) reinterpret_cast<fk_u32x4*>(stage)[idx] = pfr[k];
            }
            __syncthreads();
        }
        // ---- pass 0: u8 -> butterfly -> twiddle -> LDS, flattened over (channel, butterfly)
        float2 v0[IT0][R0];
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            int g = threadIdx.x + T * it;
            g = g < total0 ? g : total0 - 1;
            // opaque to the optimiser: otherwise the 2*R0 per-thread byte offsets (and, below, the R0 store offsets)
            // are hoisted out of the unit loop, held in registers for its whole length and partly spilled -- and
            // every spill reload waits for ALL outstanding global loads and stores (vmcnt is in order)
            asm volatile("" : "+v"(g));
            const int c = g / m0, j = g - c * m0;
            uint8_t pa[R0], pb[R0];
            bool ok[R0];
#pragma unroll
            for (int k = 0; k < R0; ++k) {                            // unconditional loads: all in flight together
                const int x = fk_reflect_src(j + k * m0, pad, cols);
                ok[k] = x >= 0;
                const int xi = (x >= 0 ? x : 0) * CH + c;
                if constexpr (STAGED) {
                    pa[k] = stage[xi];
                    pb[k] = stage[rowbytes + xi];
                } else {
                    pa[k] = row_a[xi];
                    pb[k] = row_b[xi];
                }
            }
#pragma unroll
            for (int k = 0; k < R0; ++k)
                v0[it][k] = make_float2(ok[k] ? static_cast<float>(pa[k]) : 0.f, (ok[k] && two) ? static_cast<float>(pb[k]) : 0.f);
        }
        if constexpr (STAGED) {
            __syncthreads();                                          // every byte has been read: the lines may be written
            if (u + 1 < u_end) issue_rows(u + 1);                     // in flight across all the passes of this unit
        }
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int g = threadIdx.x + T * it;
            if (g < total0) {
                const int c = g / m0, j = g - c * m0;
                Bfly<R0, false>::run(v0[it]);
                float2* zc = z + c * zs;
                zc[PL::at(j)] = v0[it][0];
#pragma unroll
                for (int q = 1; q < R0; ++q) zc[PL::at(j + q * m0)] = cmul(v0[it][q], p0w(it, q, j));
            }
        }
        __syncthreads();
        fk_inner_passes<PL, 1, CH, T, false>(z, zs, twl);
        fk_mid_lds<PL, T, CH>(z, zs, mpl);
        __syncthreads();
        fk_inner_passes<PL, P - 2, CH, T, true>(z, zs, twl);
        // ---- inverse pass 0: LDS -> conj twiddle -> butterfly -> cropped float rows
        if constexpr (STAGED) claim_rows();      // before this unit's stores (see claim_rows)
        // offsets inside a plane are 32-bit (the launcher checks plane_elems < 2^31): sixteen 64-bit per-thread
        // offsets kept across the unit loop were what spilled, and every spill reload drains the memory queue
        const int strip_step = npairs * (2 * tile_w);
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            int g = threadIdx.x + T * it;
            asm volatile("" : "+v"(g));
            if (g < total0) {
                const int c = g / m0, j = g - c * m0;
                const float2* zc = z + c * zs;
                // byte offset from the unit's (uniform) base in 32 bits: one address register per store, 24-bit multiply
                const unsigned cbase = static_cast<unsigned>(c) * static_cast<unsigned>(plane_elems);
                char* const out_b = reinterpret_cast<char*>(out_u);
                float2 v[R0];
                v[0] = zc[PL::at(j)];
#pragma unroll
                for (int q = 1; q < R0; ++q) v[q] = cmulc(zc[PL::at(j + q * m0)], p0w(it, q, j));
                Bfly<R0, true>::run(v);
#pragma unroll
                for (int k = 0; k < R0; ++k) {
                    const int x = j + k * m0 - pad;
#ifdef FK_ABL_NOSTORE   // ablation build: no global writes, values kept alive
                    asm volatile("" ::"v"(v[k].x), "v"(v[k].y), "v"(x));
                    if (false)
#else
                    if (x >= 0 && x < cols)
#endif
                    {
                        if constexpr (tile_shift != 0) {
                            const unsigned off = cbase + __umul24(static_cast<unsigned>(x) >> tile_shift, static_cast<unsigned>(strip_step)) +
                                                 2u * (static_cast<unsigned>(x) & (tile_w - 1));
                            *reinterpret_cast<float2*>(out_b + off * 4u) = v[k];
                        } else {
                            const unsigned off = cbase + static_cast<unsigned>(x);
                            *reinterpret_cast<float*>(out_b + off * 4u) = v[k].x;
                            if (two) *reinterpret_cast<float*>(out_b + (off + static_cast<unsigned>(cols)) * 4u) = v[k].y;
                        }
                    }
                }
            }
        }
    }
}

// ======================================================================================
// column pass
// ======================================================================================
template <class PL, int T, int C, int CH, bool tiled>
__global__ __launch_bounds__(T, PL::col_min_waves) void fast_colpass_u8(const float* __restrict__ planes, uint8_t* __restrict__ dst,
                                                     int rows, int cols, int pad, int nstrips, int nunits,
                                                     const float2* __restrict__ tw, const float* __restrict__ mperm)
{
    static_assert(PL::valid(), "radices do not multiply to N");
    static_assert(C % 2 == 0, "the tiled gather moves two complex lines (16 bytes) per load");
    constexpr int N = PL::N, P = PL::P, G = 2 * C;
    const float* const planes0 = planes;
    uint8_t* const dst0 = dst;
    const int npairs = (rows + 1) / 2;
    const size_t plane_elems = tiled ? static_cast<size_t>(nstrips) * npairs * (2 * G) : static_cast<size_t>(rows) * cols;
    constexpr int R0 = PL::R[0], m0 = PL::m(0);
    // pass 0: K thread groups of m0 butterflies, group gi takes lines gi, gi+K, ...
    constexpr int IT0 = Pass0Regs<PL, T>::IT;
    constexpr int K = IT0 == 1 ? (T / m0 > C ? C : T / m0) : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* z = reinterpret_cast<float2*>(smem);
    constexpr int zs = PL::zs();
    float2* twl = z + C * zs;
    float* mpl = reinterpret_cast<float*>(twl + ((PL::lds_tw_count() + 1) & ~1));         // N multipliers, position order
    uint8_t* stage = reinterpret_cast<uint8_t*>(mpl + N);                                  // [rows][G*CH] bytes

    const int gi = IT0 == 1 ? threadIdx.x / m0 : 0;
    const int j0 = IT0 == 1 ? threadIdx.x - gi * m0 : threadIdx.x;
    const bool p0_active = IT0 > 1 || gi < K;
#ifdef FK_STAMPS
    unsigned long long st_acc[kStampSlots] = {};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
    // pass 0's twiddles: registers (butterfly j0 of pass 0 is this thread's for every line), or LDS for plans on a register diet
    float2* const tw0l = reinterpret_cast<float2*>(stage + ((static_cast<size_t>(rows) * G * CH + 15) & ~static_cast<size_t>(15)));
    Pass0Regs<PL, T> p0;
    if constexpr (PL::tw0_lds) {
        for (int i = threadIdx.x; i < (R0 - 1) * m0; i += T) tw0l[i] = tw[i];
    } else {
        p0.load(tw, j0);
    }
    auto p0w = [&](int it, int q, int j) -> float2 {
        if constexpr (PL::tw0_lds) return tw0l[(q - 1) * m0 + j];
        else return p0.w[it][q];
    };
    for (int i = threadIdx.x; i < PL::lds_tw_count(); i += T) twl[i] = tw[PL::lds_tw_begin() + i];
    for (int i = threadIdx.x; i < N; i += T) mpl[i] = mperm[i];

    // Persistent workgroup.  Workgroups b, b+8, b+16, ... share an XCD and its L2 (dispatch is
    // round-robin over the 8 XCDs: speed only, never correctness).  Each XCD gets one contiguous
    // run of (frame, strip) units and its workgroups walk it INTERLEAVED, so that at any moment
    // they sit on neighbouring strips: a 128-byte line of the float planes (4 strips wide) and of
    // the u8 output (5.3 strips wide) is then touched by all its users while it is still in L2.
    // (fewer than 8 workgroups -- an image a few strips wide: as many runs as there are workgroups, or units would be
    // assigned to XCD slots nobody occupies)
    const int nx = gridDim.x < 8 ? static_cast<int>(gridDim.x) : 8;
    const int xcd = blockIdx.x % nx, lane_in_xcd = blockIdx.x / nx;
    const int wg_in_xcd = (static_cast<int>(gridDim.x) - xcd + nx - 1) / nx;
    const int u_begin = static_cast<int>(static_cast<long long>(xcd) * nunits / nx);
    const int u_end = static_cast<int>(static_cast<long long>(xcd + 1) * nunits / nx);
    // Strip layout only: the gather of task t+1 = (unit, channel) is issued into REGISTERS when
    // the passes of task t start and committed to LDS when they are done, so its ~2 us round
    // trip hides behind the FFT instead of standing in front of it.
    // Gather of one (strip, channel): for every line l (two columns) ONE 16-byte load per storage
    // row pair (two columns x two rows -> two positions of line l).  The reflected border
    // positions are then copied inside LDS from the body rows they mirror (Source.cpp:525-529),
    // the trailing zeros written; neither needs global memory.
    const int n_items = npairs * C;
    constexpr int KG = tiled ? ((N / 2) * C + T - 1) / T : 1;      // rows <= N - 2 pad  =>  npairs <= N/2
    float4 pf[KG];
    auto issue_gather = [&](int uu, int cc) {
        const int ff = uu / nstrips, ss = uu - ff * nstrips;
        const float* sb = planes0 + (static_cast<size_t>(ff) * CH + cc) * plane_elems + static_cast<size_t>(ss) * npairs * (2 * G);
        // the thread index is made opaque wherever per-thread offsets follow from it: otherwise they are hoisted out of
        // the unit loop, one or two registers each, and the kernel drowns in loop invariants (fast_rowpass3_u8: 168 -> 102 VGPRs)
        int tid = threadIdx.x;
        if constexpr ((PL::col_opaque & 1) != 0) asm volatile("" : "+v"(tid));
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            int idx = tid + T * k;
            idx = idx < n_items ? idx : n_items - 1;
            const int q = idx / C, l = idx - q * C;
#ifdef FK_ABL_NOLOAD
            pf[k] = make_float4(q * 0.5f, l * 1.f, q * 0.25f, 3.f);
#else
            pf[k] = *reinterpret_cast<const float4*>(sb + static_cast<size_t>(q) * (2 * G) + 4 * l);
#endif
        }
    };
    auto commit_gather = [&](int xx0) {
        int tid = threadIdx.x;
        if constexpr ((PL::col_opaque & 2) != 0) asm volatile("" : "+v"(tid));
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const int idx = tid + T * k;
            if (idx < n_items) {
                const int q = idx / C, l = idx - q * C;
                const int col = xx0 + 2 * l;
                float4 t = pf[k];                                 // (a_col, b_col, a_col+1, b_col+1): rows a = 2q, b = 2q+1
                if (col >= cols) { t.x = 0.f; t.y = 0.f; }        // columns beyond the image hold no data
                if (col + 1 >= cols) { t.z = 0.f; t.w = 0.f; }
                float2* zl = z + l * zs;
                zl[PL::at(pad + 2 * q)] = make_float2(t.x, t.z);
                if (2 * q + 1 < rows) zl[PL::at(pad + 2 * q + 1)] = make_float2(t.y, t.w);
            }
        }
        __syncthreads();
        // reflect-101 borders from the rows already in LDS, and the trailing zeros
        for (int idx = threadIdx.x; idx < (N - rows) * C; idx += T) {
            const int e = idx / C, l = idx - e * C;
            const int p = e < pad ? e : rows + e;                 // 0..pad-1, then pad+rows .. N-1
            const int r = fk_reflect_src(p, pad, rows);
            float2* zl = z + l * zs;
            zl[PL::at(p)] = r >= 0 ? zl[PL::at(pad + r)] : make_float2(0.f, 0.f);
        }
    };

    // "claim" = make the compiler wait for the prefetched strip at a chosen point (see fast_rowpass3_u8): before the
    // write-out stores on every path, so that the loop carries only stores across its back edge and no wait for a
    // load ever has to cover them
    auto claim_gather = [&]() {
#pragma unroll
        for (int k = 0; k < KG; ++k) asm volatile("" ::"v"(pf[k].x), "v"(pf[k].y), "v"(pf[k].z), "v"(pf[k].w));
    };
    if constexpr (tiled && FK_COL_PREFETCH) {
        if (u_begin + lane_in_xcd < u_end) issue_gather(u_begin + lane_in_xcd, 0);
        claim_gather();
    }
    for (int u = u_begin + lane_in_xcd; u < u_end; u += wg_in_xcd) {
        const int f = u / nstrips, strip = u - f * nstrips;
        planes = planes0 + static_cast<size_t>(f) * plane_elems * CH;
        dst = dst0 + static_cast<size_t>(f) * rows * cols * CH;
        const int x0 = strip * G;
        for (int ch = 0; ch < CH; ++ch) {
            const float* plane = planes + static_cast<size_t>(ch) * plane_elems;
            FK_STAMP(0);       // prologue / loop overhead / write-out of the previous strip
            __syncthreads();   // z free again (previous channel's inverse pass 0 has read it)
            FK_STAMP(3);       // barrier
            // ---- gather: line l, position p  <-  plane[reflect(p)][x0 + 2l .. +1]
            const bool full = x0 + G <= cols && (cols & 1) == 0;
            if constexpr (tiled) {
                if constexpr (!FK_COL_PREFETCH) issue_gather(u, ch);
                commit_gather(x0);
            } else if (full) {
                // unconditional 8-byte loads (clamped row, masked value): the unrolled loop keeps
                // several independent loads in flight per thread
                FK_UNROLL(FK_GATHER_UNROLL)
                for (int idx = threadIdx.x; idx < N * C; idx += T) {
                    const int p = idx / C, l = idx - p * C;
                    const int r = fk_reflect_src(p, pad, rows);
                    const float2 t = *reinterpret_cast<const float2*>(plane + static_cast<size_t>(r >= 0 ? r : 0) * cols + x0 + 2 * l);
                    z[l * zs + PL::at(p)] = r >= 0 ? t : make_float2(0.f, 0.f);
                }
            } else {
#pragma unroll 4
                for (int idx = threadIdx.x; idx < N * C; idx += T) {
                    const int p = idx / C, l = idx - p * C;
                    const int r = fk_reflect_src(p, pad, rows);
                    const int col = x0 + 2 * l;
                    const float* s = plane + static_cast<size_t>(r >= 0 ? r : 0) * cols;
                    const float a = s[col < cols ? col : cols - 1], b = s[col + 1 < cols ? col + 1 : cols - 1];
                    z[l * zs + PL::at(p)] = make_float2((r >= 0 && col < cols) ? a : 0.f, (r >= 0 && col + 1 < cols) ? b : 0.f);
                }
            }
            FK_STAMP(1);       // strip gather (global -> LDS)
            __syncthreads();
            if constexpr (tiled && FK_COL_PREFETCH) {
                const int nu = ch + 1 < CH ? u : u + wg_in_xcd, nch = ch + 1 < CH ? ch + 1 : 0;
                if (nu < u_end) issue_gather(nu, nch);
            }
            // ---- pass 0 (register twiddles)
            if (p0_active) {
#pragma unroll
                for (int it = 0; it < IT0; ++it) {
                    int j = j0 + T * it;
                    if constexpr ((PL::col_opaque & 4) != 0) asm volatile("" : "+v"(j));
                    if (j < m0) {
#pragma unroll 1
                        for (int c = gi; c < C; c += K) {
                            float2* zc = z + c * zs;
                            float2 v[R0];
#pragma unroll
                            for (int k = 0; k < R0; ++k) v[k] = zc[PL::at(j + k * m0)];
                            Bfly<R0, false>::run(v);
                            zc[PL::at(j)] = v[0];
#pragma unroll
                            for (int q = 1; q < R0; ++q) zc[PL::at(j + q * m0)] = cmul(v[q], p0w(it, q, j));
                        }
                    }
                }
            }
            __syncthreads();
            FK_STAMP(2);       // barrier + pass 0 + barrier
            fk_inner_passes<PL, 1, C, T, false>(z, zs, twl);
            FK_STAMP(4);       // forward inner passes
            fk_mid_lds<PL, T, C>(z, zs, mpl);
            __syncthreads();
            FK_STAMP(5);       // fused middle
            fk_inner_passes<PL, P - 2, C, T, true>(z, zs, twl);
            FK_STAMP(6);       // inverse inner passes
            // ---- inverse pass 0 -> "+0.5f, truncate" -> pixel stage (Utils.hpp:189,204-206)
            if (p0_active) {
#pragma unroll
                for (int it = 0; it < IT0; ++it) {
                    int j = j0 + T * it;
                    if constexpr ((PL::col_opaque & 8) != 0) asm volatile("" : "+v"(j));
                    if (j < m0) {
#pragma unroll 1
                        for (int c = gi; c < C; c += K) {
                            const float2* zc = z + c * zs;
                            float2 v[R0];
                            v[0] = zc[PL::at(j)];
#pragma unroll
                            for (int q = 1; q < R0; ++q) v[q] = cmulc(zc[PL::at(j + q * m0)], p0w(it, q, j));
                            Bfly<R0, true>::run(v);
#pragma unroll
                            for (int k = 0; k < R0; ++k) {
                                const int r = j + k * m0 - pad;
                                if (r >= 0 && r < rows) {
                                    uint8_t* s = stage + (static_cast<size_t>(r) * G + 2 * c) * CH + ch;
                                    s[0] = static_cast<uint8_t>(static_cast<int>(v[k].x + 0.5f));
                                    s[CH] = static_cast<uint8_t>(static_cast<int>(v[k].y + 0.5f));
                                }
                            }
                        }
                    }
                }
            }
            // the strip requested at the start of this task has long arrived: claim it at the end of EVERY task, so that
            // no path reaches the next commit (or the next request, which reuses the registers) with loads pending
            if constexpr (tiled && FK_COL_PREFETCH) claim_gather();
        }
        FK_STAMP(7);           // inverse pass 0 + pixel stage
        __syncthreads();
        // ---- write the strip as whole pixels: G*CH contiguous bytes per image row
        constexpr int RB = G * CH;
        if (x0 + G <= cols && ((cols * CH) & 7) == 0 && (RB & 7) == 0) {
            // 8-byte stores: three per image row of the strip
            constexpr int RQ = RB / 8;
            const uint2* s64 = reinterpret_cast<const uint2*>(stage);
            FK_UNROLL(4)
            for (int idx = threadIdx.x; idx < rows * RQ; idx += T) {
                const int r = idx / RQ, d = idx - r * RQ;
#ifdef FK_ABL_STORELOCAL   // ablation build: same store instructions, all into one small (cache-resident) region
                uint2* o = reinterpret_cast<uint2*>(dst0 + ((static_cast<size_t>(r) * 24 + blockIdx.x * 64) & 0xfff8));
#else
                uint2* o = reinterpret_cast<uint2*>(dst + (static_cast<size_t>(r) * cols + x0) * CH);
#endif
#ifdef FK_ABL_NOSTORE
                asm volatile("" ::"v"(s64[idx].x), "v"(s64[idx].y), "v"(o));
#else
                o[d] = s64[idx];
#endif
            }
        } else if (x0 + G <= cols && ((cols * CH) & 3) == 0 && (RB & 3) == 0) {
            constexpr int RD = RB / 4;
            const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
            for (int idx = threadIdx.x; idx < rows * RD; idx += T) {
                const int r = idx / RD, d = idx - r * RD;
                uint32_t* o = reinterpret_cast<uint32_t*>(dst + (static_cast<size_t>(r) * cols + x0) * CH);
                o[d] = s32[idx];
            }
        } else {
            const int wbytes = (cols - x0 < G ? cols - x0 : G) * CH;
            for (int idx = threadIdx.x; idx < rows * RB; idx += T) {
                const int r = idx / RB, b = idx - r * RB;
                if (b < wbytes) dst[(static_cast<size_t>(r) * cols + x0) * CH + b] = stage[idx];
            }
        }
    }
#ifdef FK_STAMPS
    FK_STAMP(0);
    if (threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(mperm) + N) + static_cast<size_t>(blockIdx.x) * kStampSlots;
        for (int i = 0; i < kStampSlots; ++i) o[i] = st_acc[i];
    }
#endif
}

// ---- launchers ---------------------------------------------------------------------------
template <class PL, int C> size_t fk_col_lds(int rows)
{
    return (static_cast<size_t>(C) * PL::zs() + ((PL::lds_tw_count() + 1) & ~1)) * sizeof(float2) +
           static_cast<size_t>(PL::N) * sizeof(float) + static_cast<size_t>(rows) * 2 * C * 3 + 16 +
           (PL::tw0_lds ? static_cast<size_t>(PL::R[0] - 1) * PL::m(0) * sizeof(float2) + 16 : 0);
}

// grid for `units` equal work items on `slots` resident workgroups: as many rounds as needed, all equally full
inline int fk_balanced_grid(int units, int slots)
{
    if (units <= slots) return units;
    const int rounds = (units + slots - 1) / slots;
    return (units + rounds - 1) / rounds;
}
// compute units of the current device (hipDeviceProp_t::multiProcessorCount), asked once
inline int fk_num_cus()
{
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <class PL, int T> hipError_t fk_launch_row_u8(hipStream_t st, const uint8_t* src, float* planes, int rows, int cols, int pad, int nframes, int tile_w,
                                               const float2* tw, const float* mperm)
{
    if (tile_w != 0 && tile_w != 8) return hipErrorInvalidValue;
    const int npairs = (rows + 1) / 2, nunits = npairs * nframes;
    {
        // 32-bit BYTE offsets inside a frame's three float planes, strip_step below 2^24 for the 24-bit multiply
        if (static_cast<size_t>(rows + 1) * (cols + 8) * 12 >= (static_cast<size_t>(1) << 32) || rows >= (1 << 20)) return hipErrorInvalidValue;
        const size_t lds = fk_row3_lds<PL, 3>();
        // the staged input needs 16-byte aligned rows (aligned 16-byte loads that never leave a row)
        const bool staged = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ((static_cast<size_t>(cols) * 3) & 15) == 0;
        auto kern = staged ? (tile_w ? fast_rowpass3_u8<PL, T, 3, 3, true> : fast_rowpass3_u8<PL, T, 3, 0, true>)
                           : (tile_w ? fast_rowpass3_u8<PL, T, 3, 3, false> : fast_rowpass3_u8<PL, T, 3, 0, false>);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        const int grid = fk_balanced_grid(nunits, fk_num_cus() * per_cu);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, st, src, planes, rows, cols, pad, npairs, nunits, tw, mperm);
        return hipGetLastError();
    }
}

template <class PL, int T, int C> hipError_t fk_launch_col_u8_c(hipStream_t st, const float* planes, uint8_t* dst, int rows, int cols, int pad, int nframes, int tiled,
                                                         const float2* tw, const float* mperm)
{
    const size_t lds = fk_col_lds<PL, C>(rows);
    if (tiled && C != 4) return hipErrorInvalidValue;          // the strip layout is 8 columns wide
    auto kern = (tiled && C == 4) ? fast_colpass_u8<PL, T, C, 3, (C == 4)> : fast_colpass_u8<PL, T, C, 3, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    const int nstrips = (cols + 2 * C - 1) / (2 * C), nunits = nstrips * nframes;
    // resident workgroups per CU as the runtime sees them (LDS of this image height and the kernel's registers)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const int grid = fk_balanced_grid(nunits, fk_num_cus() * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, st, planes, dst, rows, cols, pad, nstrips, nunits, tw, mperm);
    return hipGetLastError();
}

template <class PL, int T> hipError_t fk_launch_col_u8(hipStream_t st, const float* planes, uint8_t* dst, int rows, int cols, int pad, int nframes, int tiled,
                                               const float2* tw, const float* mperm, int C)
{
    switch (C) {
    case 4: return fk_launch_col_u8_c<PL, T, 4>(st, planes, dst, rows, cols, pad, nframes, tiled, tw, mperm);
    case 2: return fk_launch_col_u8_c<PL, T, 2>(st, planes, dst, rows, cols, pad, nframes, tiled, tw, mperm);
    default: return hipErrorInvalidValue;
    }
}

template <class PL> size_t fk_col_lds_bytes(int rows, int C)
{
    switch (C) {
    case 4: return fk_col_lds<PL, 4>(rows);
    case 2: return fk_col_lds<PL, 2>(rows);
    default: return ~static_cast<size_t>(0);
    }
}

// A plan serves one role: the row pass and the column pass of the same FFT length want different
// radix orders and thread counts (pass 0 should have about T butterflies; the column kernel also
// carries the prefetch registers), so each role instantiates only its own kernel.
template <class PL, int T> FastEntry fk_make_row_entry()
{
    FastEntry e{};
    e.n = PL::N;
    e.npass = PL::P;
    for (int i = 0; i < PL::P; ++i) e.radix[i] = PL::R[i];
    e.row_u8 = fk_launch_row_u8<PL, T>;
    return e;
}

template <class PL, int T> FastEntry fk_make_col_entry()
{
    FastEntry e{};
    e.n = PL::N;
    e.npass = PL::P;
    for (int i = 0; i < PL::P; ++i) e.radix[i] = PL::R[i];
    e.col_u8 = fk_launch_col_u8<PL, T>;
    e.col_lds_bytes = fk_col_lds_bytes<PL>;
    return e;
}

}  // namespace blur_amd

// one translation unit per (FFT length, role):
//   BLUR_FAST_ROW(4000, FLAGS, T, 16, 10, 5, 5)      BLUR_FAST_COL(2304, FLAGS, T, 9, 16, 16)
#define BLUR_FAST_ROW(NN, PAD, T, ...)                                                           \
    namespace blur_amd {                                                                         \
    const FastEntry* fast_row_entry_##NN()                                                       \
    {                                                                                            \
        static const FastEntry e = fk_make_row_entry<StaticPlan<NN, PAD, __VA_ARGS__>, T>();     \
        return &e;                                                                               \
    }                                                                                            \
    }
#define BLUR_FAST_COL(NN, PAD, T, ...)                                                               \
    namespace blur_amd {                                                                         \
    const FastEntry* fast_col_entry_##NN()                                                       \
    {                                                                                            \
        static const FastEntry e = fk_make_col_entry<StaticPlan<NN, PAD, __VA_ARGS__>, T>();                \
        return &e;                                                                               \
    }                                                                                            \
    }
