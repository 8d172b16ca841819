// tools/valu_rate.hip -- what does one wave64 FP32 VALU instruction cost on gfx950, scalar vs packed?
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters)
{
    float a[8]; float2v p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (float2v){ a[i], a[i] + 1.f }; }
    const float c = 1.0001f, d = 0.0001f;
    const float2v pc = { c, c }, pd = { d, d };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], c, d); }                 // 8 independent v_fma_f32
            else           { for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pc, pd); }     // 8 independent v_pk_fma_f32
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd)
{
    const int iters = 20000, blocks = 256 * waves_per_simd, threads = 256;   // 256 threads = 4 waves = 1 per SIMD
    float* d; hipMalloc(&d, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 100);
    hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = double(iters) * 64;
    const double flops = double(blocks) * threads * iters * 64 * (MODE ? 4 : 2);
    std::printf("%-12s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD, %.1f TFLOP/s\n", name, waves_per_simd, ms,
                ms * 1e6 / (instr_per_wave * waves_per_simd), flops / ms / 1e9);
    hipFree(d);
}
int main() { for (int w : { 1, 2, 4 }) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); } }
