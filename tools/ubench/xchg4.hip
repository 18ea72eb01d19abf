// Microbenchmark (diagnostic only): the SMALL all-gather of the HyperLSTM scan (hhat_t: one hyper unit x 16 batch rows
// = 64 B per workgroup, 128 workgroups per group, 2 groups).
//   V=0  counter protocol (scan_common.h): 64-B sc1 payload, drain, barrier, replicated sharded counter, poll, sc1 loads
//   V=1  tagged 8-byte granules {value, epoch} (cdna_hip_programming.md Guideline 16, R2): one 128-B line per workgroup
//        written by ONE wave instruction, no drain / barrier / counter; every wave re-reads its quarter (4 KB) until
//        every tag equals the epoch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../asr_amd/csrc/scan_common.h"
#define NG 128
#ifndef WORK
#define WORK 0
#endif
#ifndef PSLEEP
#define PSLEEP 0
#endif
__device__ __forceinline__ float val(int it, int wg, int e) { return (float)(((it * 131 + wg * 7 + e) & 0xffff) + 1); }

template <int V>
__global__ __launch_bounds__(256) void xchg(float* buf, unsigned* sync, int iters, unsigned long long* out, unsigned* bad) {
    __shared__ int flag;
    const int wg = blockIdx.x % NG, grp = blockIdx.x / NG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    float* ex = buf + (size_t)grp * 2 * NG * 32;          // [2 parities][NG][32 floats = 128 B]
    GroupSync gs;
    group_sync_init(gs, sync + grp * SYNC_STRIDE, NG, wg, &flag);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc_(ex, 2 * NG * 32 * 4);
    unsigned long long t_a = 0, t_b = 0;
    unsigned nbad = 0;
    bool giveup = false;
    group_publish(gs);
    if (!group_wait(gs)) return;
    for (int it = 0; it < iters && !giveup; ++it) {
        const int par = it & 1;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        // publish: wave 0, lanes 0..15 (batch rows)
        if (V == 0) {
            if (wave == 0 && lane < 16) st_sc1(ex + ((size_t)par * NG + wg) * 32 + lane, val(it, wg, lane));
            group_publish(gs);
        } else {
            if (wave == 0 && lane < 16) {
                unsigned long long g = ((unsigned long long)(unsigned)(it + 1) << 32) | __float_as_uint(val(it, wg, lane));
                __hip_atomic_store((unsigned long long*)(ex + ((size_t)par * NG + wg) * 32) + lane, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (V == 0) {
            if (!group_wait(gs)) break;
            // each wave reads 32 workgroups x 64 B = 2 KB: lane -> (wg = 32*wv + lane/2 ... ) as 2 b128? keep simple: b128 x 2
            f32x4 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int w_ = wv * 32 + i * 16 + (lane >> 2);            // 16 workgroups per instruction, 4 lanes x 16 B each
                a[i] = ldb_sc1_f4(rs, (unsigned)(((par * NG + w_) * 32 + (lane & 3) * 4) * 4), 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (a[i][e] != val(it, wv * 32 + i * 16 + (lane >> 2), (lane & 3) * 4 + e)) nbad++;
        } else {
            // each wave reads 32 workgroups x 128 B = 4 KB: 4 b128 per lane; 8 lanes per workgroup line
            int spins = 0;
            f32x4 a[4];
            for (;;) {
                asm volatile("" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int w_ = wv * 32 + i * 8 + (lane >> 3);
                    a[i] = ldb_sc1_f4(rs, (unsigned)(((par * NG + w_) * 32 + (lane & 7) * 4) * 4), 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) ok &= (__float_as_uint(a[i][1]) == (unsigned)(it + 1)) && (__float_as_uint(a[i][3]) == (unsigned)(it + 1));
                if (__all(ok)) break;
                if (++spins > 20000) { giveup = true; nbad += 1000000; break; }
                if (PSLEEP) __builtin_amdgcn_s_sleep(PSLEEP);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int w_ = wv * 32 + i * 8 + (lane >> 3);
                if (a[i][0] != val(it, w_, (lane & 7) * 2) || a[i][2] != val(it, w_, (lane & 7) * 2 + 1)) nbad++;
            }
        }
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        giveup = __syncthreads_or(giveup);
        if (WORK) __builtin_amdgcn_s_sleep(WORK);
        t_a += t1 - t0; t_b += t2 - t1;
    }
    if (tid == 0) { out[blockIdx.x * 2 + 0] = t_a; out[blockIdx.x * 2 + 1] = t_b; }
    if (nbad) atomicAdd(bad, nbad);
}

template <int V> void run(const char* name, int iters) {
    float* buf; unsigned* sync; unsigned long long* out; unsigned* bad;
    const size_t bb = 2 * 2 * NG * 32 * 4;
    hipMalloc(&buf, bb); hipMalloc(&sync, 2 * SYNC_STRIDE * 4); hipMalloc(&out, 256 * 2 * 8); hipMalloc(&bad, 4);
    hipFuncSetAttribute((const void*)xchg<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(buf, 0, bb); hipMemset(sync, 0, 2 * SYNC_STRIDE * 4); hipMemset(bad, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(xchg<V>, dim3(256), dim3(256), 82 * 1024, 0, buf, sync, iters, out, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 2); unsigned hb;
        hipMemcpy(h.data(), out, 256 * 2 * 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < 256; ++i) { a += h[i * 2]; b += h[i * 2 + 1]; }
        if (rep) printf("%-22s WORK=%d PSLEEP=%d %7.3f us/iter   cycles/iter: publish %6.0f receive %6.0f   mismatches %u\n", name, WORK, PSLEEP,
                        ms * 1e3 / iters, a / 256 / iters, b / 256 / iters, hb);
        fflush(stdout);
    }
}
int main() {
    run<0>("counter, 64 B/WG", 3000);
    run<1>("tagged granules", 3000);
    return 0;
}
