// Ablation timing of gemm_split_kernel (diagnostic): build with -DSPL_ABL=n, see tools/ubench/gemm_abl.sh
#include "../../asr_amd/csrc/gemm.hip"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
    // argv[1]: 0 = NN (A [M][K], B [K][N]), 1 = NT (B [N][K]), 2 = TN (A [K][M])
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int ta = mode == 2, tb = mode == 1;
    const int M = mode == 2 ? 1024 : 36192, N = 2048, K = mode == 2 ? 36192 : 1024;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 4); hipMemset(B, 0x3c, (size_t)K * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int terms = 6; terms >= 1; terms -= 5) {
        for (int r = 0; r < 3; ++r) asr_gemm_bf16x3_ws(terms, ta, tb, M, N, K, 1.f, A, ta ? M : K, B, tb ? K : N, 0.f, C, N, nullptr, nullptr, 0, 0);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) asr_gemm_bf16x3_ws(terms, ta, tb, M, N, K, 1.f, A, ta ? M : K, B, tb ? K : N, 0.f, C, N, nullptr, nullptr, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode=%d ABL=%d terms=%d  %.1f us  %.1f TFLOP/s\n",
               mode,
#ifdef SPL_ABL
               SPL_ABL,
#else
               0,
#endif
               terms, ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
