// Shared device/host helpers for the asr_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASR_OK 0
#define ASR_ERR_ARG (-1)        // bad argument / unsupported shape
#define ASR_ERR_LAUNCH (-2)     // HIP launch error
#define ASR_ERR_WS (-3)         // workspace too small
#define ASR_ERR_TIMEOUT (-4)    // an in-kernel group wait gave up (reported by asr_scan_status)
#define ASR_ERR_LIMIT (-5)      // problem larger than a fixed built-in buffer (asr_colreduce_static)

#define ASR_CHECK_LAUNCH()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return ASR_ERR_LAUNCH;   \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ASR_EPS 1.1920928955078125e-07f   // numpy.finfo(float32).eps, libs/lhuclstm_layers.py:13-14

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh via exp with full-range safety: tanh(x) = 1 - 2/(exp(2x)+1)
__device__ __forceinline__ float tanhf_(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}
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
