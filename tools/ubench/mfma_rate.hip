// Diagnostic: issue interval of v_mfma_f32_16x16x4_f32 from ONE wave per SIMD (the persistent scans' occupancy), as a
// dependent chain and as four independent chains, in s_memtime ticks and in wall-clock ns (s_memrealtime, 100 MHz).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate && tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int n) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CH], 0, 0, 0);
    }
    f32x4 s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];      // consume before the second stamp
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&t, 256 * 16);
    unsigned long long h[512];
    const int n = 256;                                  // 4096 MFMAs per wave
    for (int grid : {1, 256}) {
        for (int ch : {1, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (ch == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, t, n);
                else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, t, n);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, t, grid * 16, hipMemcpyDeviceToHost);
            printf("grid %3d chains %d: %.1f s_memtime ticks per MFMA, %.2f ns per MFMA (s_memrealtime) -> %.2f ticks/ns\n", grid, ch,
                   (double)h[0] / (16.0 * n), (double)h[1] * 10.0 / (16.0 * n), (double)h[0] / ((double)h[1] * 10.0));
        }
    }
    return 0;
}
