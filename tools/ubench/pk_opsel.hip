// Does v_pk_fma_f32 with a lo-lane operand swizzle (op_sel:[0,1,0]: the LOW result takes the HIGH half of src1)
// always give fma(a.lo, m.hi, c.lo)?  Written while looking for the cause of rare, 16-lane-wide wrong dA values in
// lstm_scan_bwd_ln2d_kernel, whose only two instructions of that form were the only two that ever went wrong
// ("a packed operation that dropped its operand swizzle").
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel tools/ubench/pk_opsel.hip && /tmp/pk_opsel [iters] [with_mfma]
//
// Prints the number of results that differ from the scalar fma, for the swizzled form and for the plain form, alone
// (with_mfma = 1) while a second stream keeps the matrix pipes of every CU busy, and (with_mfma = 2) while a second
// stream streams loads from HBM into the register files of the same SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void pk_kernel(int iters, unsigned long long* bad /* [4] */, float seed) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x2 a = {0.25f + 1e-3f * (tid & 1023), -0.5f + 2e-3f * (tid & 511)};
    f32x2 m = {seed * 1e-7f, seed * 0.37f};            // (m1, m2): m1 tiny, m2 not, as in the scan
    f32x2 c = {1e-2f * (tid & 63), -3e-2f * (tid & 31)};
    unsigned long long b_sw_lo = 0, b_sw_hi = 0, b_pl_lo = 0, b_pl_hi = 0;
    const bool odd = (threadIdx.x >> 4) & 1;           // half of the 16-lane rows sit out the swizzled form
    for (int i = 0; i < iters; ++i) {
        f32x2 d, e;
        if (!odd) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(m), "v"(c));
            const float r0 = __builtin_fmaf(a[0], m[1], c[0]), r1 = __builtin_fmaf(a[1], m[1], c[1]);
            b_sw_lo += (__float_as_uint(d[0]) != __float_as_uint(r0));
            b_sw_hi += (__float_as_uint(d[1]) != __float_as_uint(r1));
        }
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(e) : "v"(a), "v"(m), "v"(c));
        const float p0 = __builtin_fmaf(a[0], m[0], c[0]), p1 = __builtin_fmaf(a[1], m[1], c[1]);
        b_pl_lo += (__float_as_uint(e[0]) != __float_as_uint(p0));
        b_pl_hi += (__float_as_uint(e[1]) != __float_as_uint(p1));
        a[0] += 1e-4f; a[1] -= 1e-4f; c[0] += 3e-5f; m[1] += 1e-6f;
        asm volatile("" : "+v"(a), "+v"(m), "+v"(c));
    }
    if (b_sw_lo) atomicAdd(bad + 0, b_sw_lo);
    if (b_sw_hi) atomicAdd(bad + 1, b_sw_hi);
    if (b_pl_lo) atomicAdd(bad + 2, b_pl_lo);
    if (b_pl_hi) atomicAdd(bad + 3, b_pl_hi);
}

__global__ __launch_bounds__(256) void mfma_busy_kernel(int iters, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    const float x = 1e-3f * threadIdx.x;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, acc, 0, 0, 0);
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

__global__ __launch_bounds__(256) void load_busy_kernel(int iters, const f32x4* src, size_t n4, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += __builtin_nontemporal_load(src + (i0 + (size_t)u * 262144) % n4);
        i0 = (i0 + 2097152 + 256) % n4;
    }
    if (acc[0] == 123.456f) sink[1] = acc[1];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    const int with_mfma = argc > 2 ? atoi(argv[2]) : 0;
    unsigned long long* bad; float* sink;
    CHECK(hipMalloc(&bad, 4 * sizeof(*bad)));
    CHECK(hipMalloc(&sink, 16));
    CHECK(hipMemset(bad, 0, 4 * sizeof(*bad)));
    hipStream_t s0, s1;
    CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    f32x4* src = nullptr; const size_t n4 = (size_t)64 << 20;      // 1 GiB: the loads come from HBM
    if (with_mfma == 2) { CHECK(hipMalloc(&src, n4 * 16)); CHECK(hipMemset(src, 0, n4 * 16)); }
    if (with_mfma == 1) mfma_busy_kernel<<<1024, 256, 0, s1>>>(iters * 4, sink);
    if (with_mfma == 2) load_busy_kernel<<<1024, 256, 0, s1>>>(iters / 16, src, n4, sink);
    pk_kernel<<<1024, 256, 0, s0>>>(iters, bad, 1.25f);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    const double n = 1024.0 * 256 * iters;
    printf("with_mfma=%d  results per form %.3g (swizzled: half of that)\n", with_mfma, n);
    printf("swizzled op_sel:[0,1,0]  wrong lo %llu  wrong hi %llu\n", h[0], h[1]);
    printf("plain                    wrong lo %llu  wrong hi %llu\n", h[2], h[3]);
    return 0;
}
