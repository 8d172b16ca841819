// tools/valu_operand_rate.hip -- does the wave64 FP32 VALU rate on gfx950 depend on how many VGPR source operands an
// instruction reads?  (hypothesis behind the kernels' ~4 cycles per butterfly instruction; tools/valu_rate.hip's
// 2.3-cycle v_fma_f32 had ONE VGPR source and two constants.)  Developer tool, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_operand_rate.hip -o tools/valu_operand_rate.bin && tools/valu_operand_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
// 8 independent chains, 8 x 8 instructions per loop trip, forced by inline asm (the compiler cannot fold or reassociate)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE> __global__ void k(float* out, int iters, float sk)
{
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + i * 0.125f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {        // v_add_f32 v, v, v: two VGPR sources
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (MODE == 1) { // v_add_f32 v, 1.0, v: one VGPR source + inline constant
#define X(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (MODE == 2) { // v_fma_f32 v, v, v, v: three VGPR sources
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (MODE == 3) { // v_mul_f32 v, s, v: one VGPR source + SGPR
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sk));
                REP8(X)
#undef X
            } else if (MODE == 4) { // v_fmac_f32 v, v, v: dst is the third source (three VGPR reads)
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 3) & 7]));
                REP8(X)
#undef X
            } else if (MODE == 5) { // v_sub_f32 with two DIFFERENT fresh sources, result to a third register (butterfly shape)
#define X(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 5) & 7]));
                REP8(X)
#undef X
            } else if (MODE == 6) { // v_mov_b32 v, v
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else {                // v_mov_b32_dpp v, v row_shr:4 (no dependent VALU write right before: no hazard nops needed)
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd)
{
    const int iters = 10000, blocks = 256 * waves_per_simd, threads = 256;   // 256 threads = 4 waves = 1 per SIMD
    float* d; hipMalloc(&d, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 100, 1.0001f);
    hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = double(iters) * 64;
    std::printf("%-44s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
                ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
    hipFree(d);
}
int main()
{
    for (int w : { 1, 2, 3, 4 }) {
        run<0>("v_add_f32 v, v, v   (2 VGPR sources)", w);
        run<1>("v_add_f32 v, 1.0, v (1 VGPR source)", w);
        run<2>("v_fma_f32 v, v, v, v (3 VGPR sources)", w);
        run<3>("v_mul_f32 v, s, v   (1 VGPR + SGPR)", w);
        run<4>("v_fmac_f32 v, v, v  (3 VGPR reads)", w);
        run<5>("v_sub_f32 d, a, b   (2 VGPR, separate dst)", w);
        run<6>("v_mov_b32 v, v", w);
        run<7>("v_mov_b32_dpp v, v row_shr:4", w);
    }
    return 0;
}
