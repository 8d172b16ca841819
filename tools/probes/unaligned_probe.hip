// tools/probes/unaligned_probe.hip -- do dwordx3 loads / stores at byte offsets 1, 2, 3 (global and raw buffer) work on this box?
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/unaligned_probe tools/probes/unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
__global__ void probe(const uint8_t* src, uint8_t* dst_g, uint8_t* dst_b, uint8_t* dst_s, uint8_t* dst_t, int n, int shift)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    const uint32_t off = 12u * t + shift;
    if (off + 12 > (uint32_t)n) return;
    const u3 a = *reinterpret_cast<const u3*>(src + off);                 // global_load_dwordx3, unaligned
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, n, 0x00020000u);
    const u3 b = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0);
    u3* pg = reinterpret_cast<u3*>(dst_g + 12 * t);
    *pg = a;
    u3* pb = reinterpret_cast<u3*>(dst_b + 12 * t);
    *pb = b;
    *reinterpret_cast<u3*>(dst_s + off) = a;                              // unaligned store
    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(dst_t, 0, n, 0x00020000u);
    __builtin_amdgcn_raw_buffer_store_b96(a, rt, off, 0, 0);                // unaligned raw buffer store (the fused kernel's output)
}
int main()
{
    const int n = 12 * 64 * 8 + 64;
    std::vector<uint8_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = (uint8_t)(i * 7 + (i >> 8));
    uint8_t *s, *dg, *db, *ds, *dt;
    hipMalloc(&s, n); hipMalloc(&dg, n); hipMalloc(&db, n); hipMalloc(&ds, n); hipMalloc(&dt, n);
    hipMemcpy(s, h.data(), n, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(dg, 0, n); hipMemset(db, 0, n); hipMemset(ds, 0, n); hipMemset(dt, 0, n);
        probe<<<8, 64>>>(s, dg, db, ds, dt, n, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<uint8_t> g(n), b(n), st(n), bt(n);
        hipMemcpy(bt.data(), dt, n, hipMemcpyDeviceToHost);
        hipMemcpy(g.data(), dg, n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n, hipMemcpyDeviceToHost); hipMemcpy(st.data(), ds, n, hipMemcpyDeviceToHost);
        int bad_g = 0, bad_b = 0, bad_s = 0, bad_t = 0, cnt = 0;
        for (int t = 0; t < 512; ++t) {
            const int off = 12 * t + shift;
            if (off + 12 > n) continue;
            for (int k = 0; k < 12; ++k) {
                ++cnt;
                bad_g += g[12 * t + k] != h[off + k];
                bad_b += b[12 * t + k] != h[off + k];
                bad_s += st[off + k] != h[off + k];
                bad_t += bt[off + k] != h[off + k];
            }
        }
        printf("shift %d: %s  global-load mismatches %d, buffer-load mismatches %d, store mismatches %d, buffer-store mismatches %d of %d\n", shift, hipGetErrorString(e), bad_g, bad_b, bad_s, bad_t, cnt);
    }
    return 0;
}
