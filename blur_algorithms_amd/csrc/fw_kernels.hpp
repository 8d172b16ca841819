// fw_kernels.hpp -- the fused matrix-core kernel for WIDE windows (13 .. 23 blocks of 16 positions: kernel half widths 73 .. 168).
//
// fx_kernels.hpp keeps (NKB - 1) / 2 column-pass accumulator tiles per channel and a wave carries all three channels: 240 AGPRs at
// NKB = 11 and no room beyond.  Here a workgroup handles ONE channel of its strip of 128 pixel columns -- the task list has the
// channel as its fastest dimension, so the three channel tasks of a strip run at the same time on neighbouring CUs of one XCD and
// share the window's cache lines -- which leaves (NKB - 1) / 2 <= 11 tiles = 176 AGPRs.  Everything else is the structure of
// fx_kernels.hpp: window of the next step staged through LDS (binary16 subnormals straight from the bytes), row pass
// D[32 rows][32 pixels] = window x Toeplitz fragments, hand-off inside the registers (scale, quirk term, hi + lo split,
// v_permlane32_swap), sliding column-pass accumulators, emission with + 0.5f truncation.  Differences:
//   * the hi halves of the fragments live in registers (4 NKB of them), the lo halves in LDS (one ds_read_b128 per use);
//   * the output bytes of one channel are every third byte of the image: single byte stores (the three channel tasks' stores meet
//     in L2 before the lines go to memory);
//   * one product per step instead of three, and it is long (5 NKB = 115 matrix instructions at NKB = 23): the vector work of a
//     step (staging 15 groups, the hand-off, the emission) is handed out between the products in a few places instead of
//     instruction by instruction.
#pragma once
#include "fx_kernels.hpp"
#include <type_traits>

#ifndef FW_TL_REGS
#define FW_TL_REGS 12     // lo halves of the fragments kept in registers (the others are read from LDS at every use; 8 -> 12 in round 4: +1.5 % at 4K, 16 the same)
#endif

namespace blur_amd {

template <int NKB> struct FwCfg {
    static constexpr int PADA = 8 * (NKB - 2), WIN = kFxChunk + 2 * PADA, GPR = WIN / 4, PER = (GPR + 7) / 8;
    // halfs per LDS row of the window: every thread commits PER groups of 4 positions, 32 apart, without a lane mask (fx_kernels.hpp:
    // FxCfg -- a masked commit is a branch in the middle of a slice of matrix instructions), so a row holds 32 PER positions; the
    // pitch is 4 mod 8 dwords (conflict-free ds_read_b128, and ds_write_b64 with the staging rows of a 16-lane group 4 apart)
    static constexpr int fw_pitch() { int dw = 16 * PER; while ((dw & 7) != 4) ++dw; return 2 * dw; }
#ifdef FW_MASKED_COMMITS      // (rounds 3-4 A/B: lane-masked commits, the row term by threads 0 .. 31 only, the window's own pitch)
    static constexpr int PW = mx_row_pitch(NKB);
#else
    static constexpr int PW = fw_pitch();
#endif
    static constexpr int NT = (NKB - 1) / 2;                          // live accumulator tiles = steps per unrolled round
    static constexpr int BUF = 32 * PW * 2;                           // bytes of one window buffer (one channel)
    static constexpr int TLOFF = 2 * BUF;                             // lo halves of the fragments: [NKB][64 lanes] x 16 bytes
    static constexpr int QOFF = TLOFF + NKB * 64 * 16;                // qrow stage: [2 buffers][x even, odd: +q, -q][32] floats
    static constexpr int LDS = QOFF + 2 * 2 * 32 * 4;
};

template <int NKB, bool QUIRK>
__global__ __launch_bounds__(256, 1) void fw_blur_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const mx_half8* __restrict__ frags, FxGeom g,
                                                     int chunks, int tps, int nseg, int ntasks, FxQuirk qk, const uint8_t* __restrict__ strips)
{
    using C = FwCfg<NKB>;
    constexpr int PADA = C::PADA, PW = C::PW, NT = C::NT, PER = C::PER;
    constexpr int IPS = (PER + NKB - 6) / (NKB - 5);             // staging items per column-pass slot from slot 5 on
    extern __shared__ __attribute__((aligned(16))) unsigned char fw_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;

    const int nx = g.nxcd, xcd = blockIdx.x % nx, in_xcd = blockIdx.x / nx, per_xcd = (ntasks + nx - 1) / nx, task = xcd * per_xcd + in_xcd;
    if (in_xcd >= per_xcd || task >= ntasks) return;
    const int c = task % 3, xc = (task / 3) % chunks, seg = (task / (3 * chunks)) % nseg, f = task / (3 * chunks * nseg);
    const int x0 = xc * kFxChunk;
    const int tile0 = seg * tps, tile1 = min(tile0 + tps, g.ntiles);
    const uint8_t* img = src + static_cast<size_t>(f) * g.rows * g.cols * 3;
    uint8_t* out = dst + static_cast<size_t>(f) * g.rows * g.cols * 3;

    // fragments: hi halves in registers; of the lo halves the first TLR in registers too, the rest in LDS (the row pass reads a
    // window fragment per block already: with every lo half from LDS as well its two products would wait for the LDS pipe)
    constexpr int TLR = FW_TL_REGS < NKB ? FW_TL_REGS : NKB;
    mx_half8 th[NKB], tlr[TLR > 0 ? TLR : 1];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) th[kb] = frags[kb * 64 + lane];
#pragma unroll
    for (int kb = 0; kb < TLR; ++kb) tlr[kb] = frags[(NKB + kb) * 64 + lane];
    {
        mx_half8* tls = reinterpret_cast<mx_half8*>(fw_lds + C::TLOFF);
        for (int i = tid; i < NKB * 64; i += 256) tls[i] = frags[NKB * 64 + i];
    }
    const mx_half8* tlp = reinterpret_cast<const mx_half8*>(fw_lds + C::TLOFF) + lane;
    auto tlo = [&](int kb) __attribute__((always_inline)) { return kb < TLR ? tlr[kb < TLR ? kb : 0] : tlp[kb * 64]; };

    float cpos = 0.5f, cneg = 0.5f;
    if (QUIRK) {
        // the column term of this chunk's pixels in the task's channel (fx_kernels.hpp: fx_quirk_cols_tile; the window buffers are
        // not in use yet: tile and taps in buffer 0, the result in buffer 1)
        static_assert(C::BUF >= 8 * (C::WIN + 2 * C::PADA + 1 + 6) + 4 * (3 * C::WIN + 4), "fx_quirk_cols_tile's scratch fits window buffer 0");
        float* qc = reinterpret_cast<float*>(fw_lds + C::BUF);
        fx_quirk_cols_tile<1>(fw_lds, qc, qk, f, x0, c, g.cols, g.pad, tid);
        const float v = qc[32 * wave + m];
        cpos = 0.5f + v;
        cneg = 0.5f - v;
        __syncthreads();                                   // before the staging writes windows over it
    }
    const int qrows = 32 * (g.ntiles + NT);
    const double qrs = QUIRK ? static_cast<double>(qk.dr) * ((g.pad & 1) ? -1.0 : 1.0) : 0.0;

    const mx_float16 zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    mx_float16 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] = zero;
    mx_float16 arow = zero, tfin = zero;
    uint32_t hl[2][2][8];               // hand-off, two of them (the next step's is made while this step's is consumed): [hi, lo][packed row pairs], block b = entries 4 b .. 4 b + 3
    uint32_t rr[4];                     // finished tile per row group: the channel's bytes of 4 rows of the lane's pixel column

    const int s0 = tile0, s1 = tile1 + NT;
    // the window's source: the image, or (edge chunks) a strip with the mirrored pixels in place -- as in fx_kernels.hpp
    constexpr int NLEFT = fx_left_strips(PADA);                // two chunks at the left edge once the window is wider than a chunk either side
    const int sidx = xc < NLEFT ? xc : (xc >= chunks - g.nright ? NLEFT + xc - (chunks - g.nright) : -1);      // uniform
    const uint32_t pitch = sidx >= 0 ? 3u * C::WIN : 3u * static_cast<uint32_t>(g.cols);
    const uint8_t* wbase = sidx >= 0 ? strips + (static_cast<size_t>(f) * (NLEFT + g.nright) + sidx) * g.rows * (3 * C::WIN) : img + 3 * (x0 - PADA);
    const uint32_t wbytes = sidx >= 0 ? static_cast<uint32_t>(g.rows) * 3u * C::WIN : (static_cast<uint32_t>(g.rows) * g.cols - static_cast<uint32_t>(x0 - PADA)) * 3u;
    const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wbase), 0, wbytes, kMxRsrcWord3);
#ifdef FW_STAGE_LINEAR
    const int srow = tid >> 3, g0 = tid & 7;
#else
    const int srow = 8 * (tid >> 6) + ((tid >> 4) & 3) + 4 * ((tid >> 3) & 1), g0 = tid & 7;      // (fx_kernels.hpp: the staging map)
#endif
    uint32_t raw[PER][3];
    int qpart[kFxMaxBatches] = {};
    // the window of step s: thread t moves the twelve-byte groups (t & 7) + 8 k of row t >> 3, all requested at once (consumed one
    // matrix-heavy pass later).  (A mapping with 2 rows x 384 contiguous bytes per wave load instead of 8 x 96 changed nothing.)
    auto issue_window = [&](int s) __attribute__((always_inline)) {
#ifdef FW_ABL_NOLOAD
        if (s > s0 + 1) return;
#endif
        const int r = mx_refl(32 * s - PADA + srow, g.rows);
        const uint32_t off = static_cast<uint32_t>(r) * pitch + 12u * g0;
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const bool in = (C::GPR % 8 == 0) || k < PER - 1 || g0 < C::GPR % 8;
            const u3 t = __builtin_amdgcn_raw_buffer_load_b96(rimg, in ? off + 96u * k : off, 0, 0);
            raw[k][0] = t[0]; raw[k][1] = t[1]; raw[k][2] = t[2];
        }
#ifdef FW_MASKED_COMMITS
        if (QUIRK && tid < 32)
#else
        if (QUIRK)                      // the row term of row re of V (= image row refl(re - PADA)): dr (-1)^pad Srow; every thread (row tid & 31:
#endif
        {                               // eight copies of each value) -- a test on tid would be a branch in the middle of a slice of matrix instructions
            const int* sp = qk.srow_part + (static_cast<size_t>(f) * qk.nbatches * g.rows + mx_refl(min(32 * s + (tid & 31), qrows - 1) - PADA, g.rows)) * 3 + c;
#pragma unroll
            for (int b = 0; b < kFxMaxBatches; ++b) qpart[b] = sp[static_cast<size_t>(min(b, qk.nbatches - 1)) * g.rows * 3];      // (fx_kernels.hpp: issue_chunk)
        }
    };
    // channel c of group k -> binary16 subnormals -> LDS: two v_perm_b32 (run-time selectors: the channel is the task's) and one
    // ds_write_b64.  Pixels (0, 1) of the group are bytes (c, 3 + c), pixels (2, 3) bytes (6 + c, 9 + c) of its three dwords.
    const uint32_t selA = c == 0 ? 0x0c030c00u : (c == 1 ? 0x0c040c01u : 0x0c050c02u);      // operands (d1, d0)
    const uint32_t selB = c == 0 ? 0x0c050c02u : (c == 1 ? 0x0c060c03u : 0x0c070c00u);      // operands (d2, d1) / c == 2: (d2, d2)
    auto commit_item = [&](int buf, int k) __attribute__((always_inline)) {
#ifdef FW_ABL_NOCOMMIT
        if (buf >= 0) return;
#endif
        if (k >= PER) return;
#ifdef FW_MASKED_COMMITS
        if ((C::GPR % 8 == 0) || k < PER - 1 || g0 < C::GPR % 8)
#endif
        {   // (groups past the window's GPR of the last k hold whatever their clamped load returned: they land in the row's padding)
            _Float16* base = reinterpret_cast<_Float16*>(fw_lds + buf * C::BUF) + srow * PW + 4 * g0;
            uint2 wd;
            wd.x = __builtin_amdgcn_perm(raw[k][1], raw[k][0], selA);
            wd.y = __builtin_amdgcn_perm(raw[k][2], c == 2 ? raw[k][2] : raw[k][1], selB);
            *reinterpret_cast<uint2*>(base + 32 * k) = wd;
        }
    };
    auto commit_q = [&](int buf) __attribute__((always_inline)) {
#ifdef FW_MASKED_COMMITS
        if (QUIRK && tid < 32) {
#else
        if (QUIRK) {                     // (every thread: the eight threads of a row store the same value to the same place)
#endif
            int v = qpart[0];
#pragma unroll
            for (int b = 1; b < kFxMaxBatches; ++b) v += b < qk.nbatches ? qpart[b] : 0;
            const float qraw = static_cast<float>(qrs * v);
            float* qs = reinterpret_cast<float*>(fw_lds + C::QOFF) + buf * 64 + (tid & 31);
            qs[0] = qraw;            // the term enters as qrow (-1)^x: lanes of even x read this copy,
            qs[32] = -qraw;          // lanes of odd x this one
        }
    };
    // R: the window in buffer `buf` -> arow; `beside(kb)` runs after the two products of window block kb
    auto rowpass = [&](int buf, auto beside) __attribute__((always_inline)) {
        const _Float16* base = reinterpret_cast<const _Float16*>(fw_lds + buf * C::BUF) + m * PW + wave * 32 + 8 * h;
        mx_float16 a = zero;
        mx_half8 x[4], tq[3];                      // window fragments three blocks ahead, LDS-resident lo halves two blocks ahead
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) x[kb] = *reinterpret_cast<const mx_half8*>(base + 16 * kb);
        tq[0] = tlo(0);
        tq[1] = tlo(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 3 < NKB) x[(kb + 3) & 3] = *reinterpret_cast<const mx_half8*>(base + 16 * (kb + 3));
            if (kb + 2 < NKB) tq[(kb + 2) % 3] = tlo(kb + 2);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kb & 3], th[kb], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kb & 3], tq[kb % 3], a, 0, 0, 0);
            asm volatile("" : "+a"(a));          // pins the two products between this block's fences
            beside(kb);
#ifndef FW_NOSB
            __builtin_amdgcn_sched_barrier(0);   // the vector work handed out beside block kb stays beside block kb
#endif
        }
        arow = a;
    };
    // S: arow -> scale (+ quirk), split into hi + lo, exchange with lane ^ 32 -> hl[hb] (fx_kernels.hpp: split_piece), in eight pieces:
    // rows 0..15 / 16..31 of the tile x {read + scale, convert row pairs 0 1, convert row pairs 2 3, exchange}
    float sv[16];
    auto split_piece = [&](int buf, int hb, int piece) __attribute__((always_inline)) {
        const int hf = piece >> 2, sub = piece & 3;
        uint32_t (&hp)[8] = hl[hb][0];
        uint32_t (&lp)[8] = hl[hb][1];
        if (sub == 0) {
            if (QUIRK) {
                const float* qs4 = reinterpret_cast<const float*>(fw_lds + C::QOFF) + buf * 64 + (m & 1) * 32 + 4 * h;
#pragma unroll
                for (int k = 2 * hf; k < 2 * hf + 2; ++k) {
                    const float4 t4 = *reinterpret_cast<const float4*>(qs4 + 8 * k);
                    sv[4 * k] = __builtin_fmaf(arow[4 * k], kFxRowUnscale, t4.x);
                    sv[4 * k + 1] = __builtin_fmaf(arow[4 * k + 1], kFxRowUnscale, t4.y);
                    sv[4 * k + 2] = __builtin_fmaf(arow[4 * k + 2], kFxRowUnscale, t4.z);
                    sv[4 * k + 3] = __builtin_fmaf(arow[4 * k + 3], kFxRowUnscale, t4.w);
                }
            } else {
#pragma unroll
                for (int k = 8 * hf; k < 8 * hf + 8; ++k) sv[k] = arow[k] * kFxRowUnscale;
            }
        } else if (sub == 1 || sub == 2) {
#pragma unroll
            for (int k = 4 * hf + 2 * (sub - 1); k < 4 * hf + 2 * sub; ++k) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const f2 vv = { sv[2 * k], sv[2 * k + 1] };
                hp[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(vv, h2));
                float r0, r1;
                mx_remainder(hp[k], vv[0], vv[1], r0, r1);
                const f2 rem = { r0, r1 };
                lp[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, h2));
            }
        } else {
            // regs (0..3, 4..7) of the half = rows (0..3, 8..11) + 4 h of its 16-row block -> the lane wants rows 8 h .. 8 h + 7
            fx_swap4(hp[4 * hf], hp[4 * hf + 2], hp[4 * hf + 1], hp[4 * hf + 3], lp[4 * hf], lp[4 * hf + 2], lp[4 * hf + 1], lp[4 * hf + 3]);
        }
    };
    // E: four registers of the finished tile (rows 8 gq + 4 h + 0 .. 3 of the lane's pixel column) -> bytes, kept in rr[gq].  No
    // transposes: the lane keeps its COLUMN, so a store instruction covers 2 rows x 32 pixels (96-byte spans, two or three cache
    // lines) instead of 8 rows x 8 quads.
    auto emit_piece = [&](int gq) __attribute__((always_inline)) {
        float fv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int reg = 4 * gq + k;
            fv[k] = __builtin_fmaf(tfin[reg], kMxUnscale, (reg & 1) ? cneg : cpos);
        }
        // (uint8_t)(v + 0.5f) of the reference (Utils.hpp:189,204-206): truncate, keep the low byte
        const uint32_t b0 = static_cast<uint32_t>(static_cast<int>(fv[0])) & 0xffu, b1 = static_cast<uint32_t>(static_cast<int>(fv[1])) & 0xffu;
        const uint32_t b2 = static_cast<uint32_t>(static_cast<int>(fv[2])) & 0xffu, b3 = static_cast<uint32_t>(static_cast<int>(fv[3]));
        rr[gq] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    };
    // C: column pass of step slot qs from hl (fx_kernels.hpp: colpass): first the tile that FINISHES, last the tile that STARTS
    // `ri`: step ri of the segment's first NT (-1: a later step) -- the triples of tiles above the segment (a2 > ri) are left out
    // statically (fx_kernels.hpp: colpass)
    auto colpass = [&](int qs, int hb, int ri, auto beside) __attribute__((always_inline)) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        mx_half8 v1[2], v2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const u4 w1 = { hl[hb][0][4 * b], hl[hb][0][4 * b + 1], hl[hb][0][4 * b + 2], hl[hb][0][4 * b + 3] };
            const u4 w2 = { hl[hb][1][4 * b], hl[hb][1][4 * b + 1], hl[hb][1][4 * b + 2], hl[hb][1][4 * b + 3] };
            v1[b] = __builtin_bit_cast(mx_half8, w1);
            v2[b] = __builtin_bit_cast(mx_half8, w2);
        }
        auto dof = [](int it) { return it == 0 ? NKB - 1 : (it >= NKB - 2 ? it - (NKB - 2) : it + 1); };
        mx_half8 tq[3];
        tq[0] = tlo(dof(0));
        tq[1] = tlo(dof(1));
#pragma unroll
        for (int it = 0; it < NKB; ++it) {
            const int d = dof(it);
            const int b = d & 1, a2 = d >> 1, slot = (qs - a2 + 2 * NT) % NT;
            if (it + 2 < NKB) tq[(it + 2) % 3] = tlo(dof(it + 2));
            if (ri < 0 || a2 <= ri) {
                mx_float16 t = d == 0 ? zero : acc[slot];
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[d], v1[b], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(tq[it % 3], v1[b], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[d], v2[b], t, 0, 0, 0);
                asm volatile("" : "+a"(t));      // pins the three products between this triple's fences
                if (it == 0) tfin = t; else acc[slot] = t;
            }
            beside(it);
#ifndef FW_NOSB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    // F: the channel's bytes of the finished tile: byte c of every pixel (stride 3), rows 8 gq + 4 h + k of the lane's column.  Buffer
    // stores: rows past the image, lanes right of it and tiles that do not exist get an offset outside the resource.
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, static_cast<uint32_t>(g.rows) * g.cols * 3u, kMxRsrcWord3);
    const int xcol = x0 + 32 * wave + m;
    const uint32_t lane_out = (static_cast<uint32_t>(4 * h) * g.cols + static_cast<uint32_t>(xcol)) * 3u + static_cast<uint32_t>(c);
    const uint32_t rowstep = static_cast<uint32_t>(g.cols) * 3u;
    auto store_group = [&](int tile, bool valid, int gq) __attribute__((always_inline)) {
        const int row0 = 32 * tile + 8 * gq + 4 * h;
        const uint32_t base = lane_out + static_cast<uint32_t>(32 * tile + 8 * gq) * rowstep;
        const uint32_t v = rr[gq];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#ifdef FW_ABL_NOSTORE
            const bool ok = false && valid;
#else
            const bool ok = valid && xcol < g.cols && row0 + k < g.rows;
#endif
            __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(v >> (8 * k)), rout, ok ? base + k * rowstep : 0xfffffff0u, 0, 0);
        }
    };

    // prologue: windows s0 and s0 + 1 in LDS, the first row pass and its hand-off done
    issue_window(s0);
#pragma unroll
    for (int k = 0; k < PER; ++k) commit_item(0, k);
    commit_q(0);
    issue_window(s0 + 1);
#pragma unroll
    for (int k = 0; k < PER; ++k) commit_item(1, k);
    commit_q(1);
    __syncthreads();
    rowpass(0, [](int) {});
#pragma unroll
    for (int p = 0; p < 8; ++p) split_piece(0, 0, p);
    __syncthreads();                                           // window s0 may be overwritten (row pass s0 has read it)

    // Step s: the row pass of step s + 1 (window s + 1), then the column pass of step s.  The step's vector work rides beside the
    // products, a few instructions per slot (a slot = the two products of a window block / the three of a column-pass triple; a
    // block of vector instructions longer than a slot's matrix time would leave the matrix pipe idle):
    //   row pass    the stores of the tile the PREVIOUS step finished (slots 0 .. 3); the loads of window s + 2 are requested before it
    //   column pass the hand-off of step s + 1 (slots 0 .. 7, into hl[1]; copied to hl[0] at the end of the step), the emission of
    //               the tile this step finishes (slots 1 .. 4), window s + 2 group by group into the buffer window s left (from slot 5)
    // The step loop is unrolled NT times: the accumulator rotation is static, the window buffers alternate through a run-time offset.
    auto step = [&](int s, int qs, int ri) __attribute__((always_inline)) {
        {
            const int par = (s - s0) & 1;                      // window s: buffer par; window s + 1: the other one
            const int ptile = s - 1 - NT;
            const bool pvalid = ptile >= tile0 && s > s0;
            issue_window(s + 2);
            rowpass(par ^ 1, [&](int kb) __attribute__((always_inline)) {
                if (kb < 4) store_group(ptile, pvalid, kb);
            });
            colpass(qs, 0, ri, [&](int it) __attribute__((always_inline)) {
                if (it < 8) split_piece(par ^ 1, 1, it);
                if (it >= 1 && it <= 4) emit_piece(it - 1);
                if (it >= 5) {
#pragma unroll
                    for (int i = 0; i < IPS; ++i) commit_item(par, IPS * (it - 5) + i);
                }
                if (it == NKB - 1) commit_q(par);
            });
#pragma unroll
            for (int k = 0; k < 8; ++k) { hl[0][0][k] = hl[1][0][k]; hl[0][1][k] = hl[1][1][k]; }
#ifndef FW_ABL_NOBARRIER
            __syncthreads();                                   // window s + 2 complete, window s + 1 no longer read
#endif
        }
    };
    // the first NT steps of a segment as their own copy of the body, without the column products of the tiles above the segment (about
    // half the column pass of these steps: NT = 11 for the widest window, where a single image's segments are 14 .. 20 steps long)
#ifndef FW_NO_PEEL
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
    {
        const int ltile = s1 - 1 - NT;                         // the tile the last step finished
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) store_group(ltile, ltile >= tile0, gq);
    }
}

template <int NKB> hipError_t fw_launch_u8(hipStream_t st, const uint8_t* src, uint8_t* dst, const void* frags, FxGeom g, int num_cus, const FxQuirk* qk,
                                           const uint8_t* strips, float* vdump, unsigned long long* stamps)
{
    using C = FwCfg<NKB>;
    if (vdump || stamps) return hipErrorNotSupported;            // the row-pass dump and the phase stamps are builds of fx_blur_u8 only
    const int chunks = (g.cols + kFxChunk - 1) / kFxChunk;
    const long long nstripes = static_cast<long long>(chunks) * g.nframes * 3;      // (strip of columns, channel)
    if (nstripes <= 0) return hipSuccess;
    // segments per strip as in fx_launch_u8: the shortest makespan = rounds x (tiles per segment + NT of run-in)
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
        const void* kernels[2] = { reinterpret_cast<const void*>(fw_blur_u8<NKB, true>), reinterpret_cast<const void*>(fw_blur_u8<NKB, false>) };
        for (const void* k : kernels) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
            if (e != hipSuccess) return e;
        }
        fx_attr_mark(attr_done, dev);
    }
    if (qk)
        hipLaunchKernelGGL((fw_blur_u8<NKB, true>), grid, dim3(256), C::LDS, st, src, dst, static_cast<const mx_half8*>(frags), g, chunks, tps, nseg,
                           static_cast<int>(ntasks), *qk, strips);
    else
        hipLaunchKernelGGL((fw_blur_u8<NKB, false>), grid, dim3(256), C::LDS, st, src, dst, static_cast<const mx_half8*>(frags), g, chunks, tps, nseg,
                           static_cast<int>(ntasks), FxQuirk{}, strips);
    return hipGetLastError();
}

#define BLUR_FW(NKB_)                                                                                       \
    namespace blur_amd {                                                                                    \
    const FxEntry* fx_entry_##NKB_()                                                                        \
    {                                                                                                       \
        static const FxEntry e = { NKB_, fw_launch_u8<NKB_> };                                              \
        return &e;                                                                                          \
    }                                                                                                       \
    }

}  // namespace blur_amd
