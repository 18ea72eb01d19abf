// Shared device/host helpers for the asr_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASR_OK 0
#define ASR_ERR_ARG (-1)        // bad argument / unsupported shape
#define ASR_ERR_LAUNCH (-2)     // HIP launch error
#define ASR_ERR_WS (-3)         // workspace too small
#define ASR_ERR_TIMEOUT (-4)    // an in-kernel group wait gave up (reported by asr_scan_status)
#define ASR_ERR_LIMIT (-5)      // larger than a fixed capacity (asr_colreduce_static's buffer; a scan's resident weights / grid)

#define ASR_CHECK_LAUNCH()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return ASR_ERR_LAUNCH;   \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ASR_EPS 1.1920928955078125e-07f   // numpy.finfo(float32).eps, libs/lhuclstm_layers.py:13-14

// ---- activations of the recurrences (T.nnet.sigmoid / T.tanh, libs/lhuclstm_layers.py:93-108) ----
// The reference evaluates them through the C library (libm-grade, relative accuracy everywhere).  The
// hardware's v_exp_f32 / v_rcp_f32 are 1 ulp each, but exp(x) = exp2(x*log2e) loses |x|*2^-24 in the
// rounded argument and tanh = 1 - 2/(e^{2x}+1) is only ABSOLUTELY accurate near 0; both showed in the
// per-layer parity numbers (profiles/parity_r03.json against parity_r02.json).  Forms used here
// (max error measured over 2^32 arguments by tests/test_gpu_kernels.py::test_activation_accuracy):
//   exp  : argument product in two floats (Cody-Waite on log2e), v_exp_f32, first-order correction
//   1/d  : v_rcp_f32 + one Newton step (skipped for d = inf, where -inf*0 would make a NaN)
//   tanh : |x| < 0.625 odd minimax polynomial (relative 4e-9 before rounding), else 1 - 2/(e^{2|x|}+1)
// ASR_ACT = 0 (old fast forms) and 2 (ocml expf / tanhf, IEEE division) exist for the cost / parity
// comparison builds of tools/act_variants.sh only; the product is built with the default.
#ifndef ASR_ACT
#define ASR_ACT 1
#endif
__device__ __forceinline__ float exp_acc_(float x) {
    const float L2E_HI = 0x1.715476p+0f, L2E_LO = 0x1.4ae0c0p-26f, LN2 = 0x1.62e430p-1f;
    const float hi = x * L2E_HI;
    const float lo = __builtin_fmaf(x, L2E_HI, -hi) + x * L2E_LO;        // x*log2(e) - hi, exact to first order
    const float e = __builtin_amdgcn_exp2f(hi);
    // 2^(hi+lo) = 2^hi (1 + lo ln2); e = +inf or 0 (overflow, underflow, x = +-inf, where lo is NaN) stays as it is
    return __builtin_amdgcn_class(e, 0x260) ? e : __builtin_fmaf(e, lo * LN2, e);
}
__device__ __forceinline__ float rcp_acc_(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float rr = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_amdgcn_class(d, 0x264) ? r : rr;                     // d = +-inf, +-0: -d*r is NaN
}
#if ASR_ACT == 1
__device__ __forceinline__ float expf_(float x) { return exp_acc_(x); }       // softmax (loss_opt.hip)
__device__ __forceinline__ float sigmoidf_(float x) { return rcp_acc_(1.0f + exp_acc_(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float y = __builtin_fabsf(x);
    const float z = y * y;
    float p = -0x1.76ac0cp-8f;
    p = __builtin_fmaf(p, z, 0x1.5256d4p-6f);
    p = __builtin_fmaf(p, z, -0x1.b84448p-5f);
    p = __builtin_fmaf(p, z, 0x1.110768p-3f);
    p = __builtin_fmaf(p, z, -0x1.555534p-2f);
    const float small = __builtin_fmaf(y * z, p, y);
    const float big = __builtin_fmaf(-2.0f, rcp_acc_(exp_acc_(2.0f * y) + 1.0f), 1.0f);
    return __builtin_copysignf(y < 0.625f ? small : big, x);
}
#elif ASR_ACT == 2
__device__ __forceinline__ float expf_(float x) { return expf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float expf_(float x) { return __expf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}
#endif
__device__ __forceinline__ float clipf_(float g, float c) {
    return c > 0.0f ? fminf(fmaxf(g, -c), c) : g;
}

// ---- agent-scope (sc1) accesses for data exchanged between workgroups inside one launch ----
// (cdna_hip_programming.md Guideline 16: payload stored write-through, every load of it sc1.)
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte sc1 load/store through a raw buffer descriptor (aux = 16 -> sc1).
typedef int i32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_ make_rsrc(const void* base, unsigned bytes) {
    uint64_t a = (uint64_t)base;
    i32x4_ r;
    r.x = (int)(a & 0xffffffffu);
    r.y = (int)((a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;   // raw buffer, dword format (DATA_FORMAT=32)
    return r;
}
