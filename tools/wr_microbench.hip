// tools/wr_microbench.hip -- what does the wave-resident middle section (wr_kernels.hpp) cost by itself, per wave task,
// at 1..4 waves per SIMD, and what do its parts cost?  Developer tool (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iblur_algorithms_amd/csrc tools/wr_microbench.hip -o tools/wr_microbench.bin && tools/wr_microbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wr_kernels.hpp"
using namespace blur_amd;

// MODE 0: whole middle section (LDS read, 4 butterflies, twiddles, multiply, 2 transposes, LDS write)
//      1: the same without the transposes      2: only the two transposes      3: only LDS read + write
//      4: only the four radix-16 butterflies    5: the 15 + 15 twiddle multiplies and the 16 scalings
template <int MODE, bool TWL> __global__ __launch_bounds__(1024) void mid(float* out, const float2* w256, const float* mult, int iters, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lines = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x, T = blockDim.x;
    float2* twl = lines + (T / 16) * kWrSB;
    wr_twl_fill(twl, w256, tid, T);
    for (int i = tid; i < (T / 16) * kWrSB; i += T) lines[i] = make_float2(0.001f * (i % 97), 0.002f * (i % 89));
    const int lane16 = (tid >> 2) & 15, sbi = (tid >> 6) * 4 + (tid & 3);
    WrMid<TWL> wm;
    wr_mid_load<16>(wm, lane16, sbi % 16, w256, mult, twl);
    __syncthreads();
    float2* sb = lines + sbi * kWrSB;
    float2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sb[16 * k + lane16];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int l16 = wr_opaque(lane16);
        if constexpr (MODE == 0) wr_middle(sb, wm, l16);
        else if constexpr (MODE == 6) { wr_middle(sb, wm, l16); __syncthreads(); }
        else if constexpr (MODE == 7) { wr_middle(sb, wm, l16); __syncthreads(); wr_middle(sb, wm, l16); __syncthreads(); wr_middle(sb, wm, l16); __syncthreads(); }
        else {
            if constexpr (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = sb[16 * k + l16];
            }
            if constexpr (MODE == 1 || MODE == 4) { Bfly<16, false>::run(v); }
            if constexpr (MODE == 1 || MODE == 5) {
#pragma unroll
                for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], wm.w(k, l16));
            }
            if constexpr (MODE == 2) wr_transpose16(v);
            if constexpr (MODE == 1 || MODE == 4) { Bfly<16, false>::run(v); }
            if constexpr (MODE == 1 || MODE == 5) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = cscale(v[r], wm.mm[r]);
            }
            if constexpr (MODE == 1 || MODE == 4) { Bfly<16, true>::run(v); }
            if constexpr (MODE == 1 || MODE == 5) {
#pragma unroll
                for (int r = 1; r < 16; ++r) v[r] = cmulc(v[r], wm.w(r, l16));
            }
            if constexpr (MODE == 2) wr_transpose16(v);
            if constexpr (MODE == 1 || MODE == 4) { Bfly<16, true>::run(v); }
            if constexpr (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 16; ++k) sb[16 * k + l16] = v[k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { asm volatile("" : "+v"(v[k].x), "+v"(v[k].y)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k].x + v[k].y + sb[16 * k + lane16].x;
    out[blockIdx.x * T + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * (T / 64) + (tid >> 6)] = t1 - t0;
}

template <int MODE, bool TWL> void run(const char* name, int waves_per_simd, const float2* w256, const float* mult)
{
    const int T = 256 * waves_per_simd, blocks = 256, iters = 400;
    float* d; hipMalloc(&d, sizeof(float) * blocks * T);
    unsigned long long* c; hipMalloc(&c, 8 * blocks * (T / 64));
    const size_t lds = 100 * 1024;      // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(mid<MODE, TWL>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mid<MODE, TWL><<<blocks, T, lds>>>(d, w256, mult, 10, c);
    hipEventRecord(e0); mid<MODE, TWL><<<blocks, T, lds>>>(d, w256, mult, iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (T / 64));
    hipMemcpy(h.data(), c, 8 * h.size(), hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += double(x); mean /= h.size();
    std::printf("%-34s %s waves/SIMD %d: %8.0f cycles per task per wave, %7.0f per task per SIMD   (%.3f ms)\n", name, TWL ? "twl" : "reg", waves_per_simd,
                mean / iters, mean / iters / waves_per_simd, ms);
    hipFree(d); hipFree(c);
}

int main()
{
    std::vector<float> w(512), m(4096, 1.f / 4096);
    for (int i = 0; i < 256; ++i) { w[2 * i] = std::cos(-6.283185307179586 * i / 256); w[2 * i + 1] = std::sin(-6.283185307179586 * i / 256); }
    float2* w256; float* mult;
    hipMalloc(&w256, 2048); hipMalloc(&mult, 4096 * 4);
    hipMemcpy(w256, w.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(mult, m.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int wps : { 1, 2, 3, 4 }) {
        run<0, false>("whole middle", wps, w256, mult);
        run<0, true>("whole middle", wps, w256, mult);
    }
    run<6, false>("whole middle + barrier", 3, w256, mult);
    run<6, true>("whole middle + barrier", 3, w256, mult);
    run<7, false>("3 x (whole middle + barrier)", 3, w256, mult);
    for (int wps : { 1, 3 }) {
        run<1, false>("no transposes", wps, w256, mult);
        run<2, false>("two transposes only", wps, w256, mult);
        run<3, false>("LDS read + write only", wps, w256, mult);
        run<4, false>("four radix-16 butterflies only", wps, w256, mult);
        run<5, false>("twiddles + scaling only", wps, w256, mult);
    }
    return 0;
}
