// Diagnostic: where the cycles of one K-tile iteration of gemm_split_kernel go (s_memtime stamps per wave, summed).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEMM_STAMPS -I asr_amd/csrc tools/ubench/gemm_stamps.hip -o tools/ubench/bin/gemm_stamps
#define GEMM_STAMPS 1
#include "../../asr_amd/csrc/gemm.hip"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;        // 0 NN, 1 NT, 2 TN
    const int ta = mode == 2, tb = mode == 1;
    const int M = mode == 2 ? 1024 : 36192, N = 2048, K = mode == 2 ? 36192 : 1024;
    float *A, *B, *C, *ws;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&ws, (size_t)128 << 20);
    hipMemset(A, 0x3c, (size_t)M * K * 4); hipMemset(B, 0x3c, (size_t)K * N * 4);
    for (int r = 0; r < 2; ++r) asr_gemm_bf16x3_ws(6, ta, tb, M, N, K, 1.f, A, ta ? M : K, B, tb ? K : N, 0.f, C, N, nullptr, ws, (size_t)128 << 20, 0);
    hipDeviceSynchronize();
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h[8];
    hipMemcpyToSymbol(HIP_SYMBOL(gemm_stamp_acc), z, sizeof(z));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    asr_gemm_bf16x3_ws(6, ta, tb, M, N, K, 1.f, A, ta ? M : K, B, tb ? K : N, 0.f, C, N, nullptr, ws, (size_t)128 << 20, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(h, HIP_SYMBOL(gemm_stamp_acc), sizeof(h));
    const double it = (double)h[5] * 4.0;                  // iterations x waves
    printf("mode=%d  %.1f us (stamped build)  per K-tile iteration and wave, s_memtime ticks: issue loads %.0f | frag reads + 24 MFMA issue %.0f | "
           "wait for the global loads %.0f | split + LDS stores %.0f | barrier %.0f   (sum %.0f)\n", mode, ms * 1e3,
           h[0] / it, h[1] / it, h[2] / it, h[3] / it, h[4] / it, (h[0] + h[1] + h[2] + h[3] + h[4]) / it);
    return 0;
}
