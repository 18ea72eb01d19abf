// Microbenchmark (diagnostic only): what a chip-wide all-gather step of the persistent scans is made of.
// 2 groups x NG workgroups (one per CU); per iteration NP of a group's workgroups publish 256 B each with the
// counter protocol of scan_common.h, then NR of them read the whole payload (the rest only wait for the counter).
// Stamps: drain (sc1 store -> vmcnt(0)), arrive (barrier + atomic issue), poll (until the counter matches),
// load (8 b128 per lane of the payload).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../asr_amd/csrc/scan_common.h"

#ifndef NG
#define NG 128
#endif
#ifndef NR
#define NR NG
#endif
#ifndef WORK
#define WORK 0
#endif
#ifndef RDELAY
#define RDELAY 0       // non-critical readers (wg >= NR) sleep this long before reading (they DO read if LATE=1)
#endif
#ifndef LATE
#define LATE 0
#endif

__global__ __launch_bounds__(256) void xchg(float* buf, unsigned* sync, int iters, unsigned long long* out) {
    __shared__ int flag;
    const int wg = blockIdx.x % NG, grp = blockIdx.x / NG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int q = lane >> 4, bl = lane & 15;
    float* rex = buf + (size_t)grp * 2 * 128 * 64;
    GroupSync gs;
    group_sync_init(gs, sync + grp * SYNC_STRIDE, NG, wg, &flag);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc_(rex, 2 * 128 * 64 * 4);
    const unsigned ex_voff = (unsigned)((q * 64 + bl * 4) * 4);
    unsigned long long t_d = 0, t_a = 0, t_p = 0, t_l = 0;
    float sink = 0.f;
    const int nblk4 = NG / 16;       // b128 groups per wave (NG blocks / 4 per b128-group-of-q / 4 waves)
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (wave == 0 && q == 0) {
            f32x4 v0 = {(float)it, (float)wg, (float)bl, 1.f};
            st_sc1_f4(rs, (unsigned)(((par * 128 + wg) * 64 + bl * 4) * 4), v0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (threadIdx.x < SYNC_REPL)
            __hip_atomic_fetch_add(gs.words + (threadIdx.x * SYNC_SHARDS + gs.my_shard) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (!group_wait(gs)) break;
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        if (wg < NR || LATE) {
            if (wg >= NR && RDELAY) __builtin_amdgcn_s_sleep(RDELAY);
            f32x4 bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                bv[i] = ldb_sc1_f4(rs, ex_voff, (unsigned)(par * 128 * 256) + (unsigned)((wv * nblk4 + (i % nblk4))) * 1024u);
#pragma unroll
            for (int i = 0; i < 8; ++i) sink += bv[i][0] + bv[i][3];
            asm volatile("" :: "v"(sink));
        }
        unsigned long long t4 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (WORK) __builtin_amdgcn_s_sleep(WORK);
        t_d += t1 - t0; t_a += t2 - t1; t_p += t3 - t2; t_l += t4 - t3;
    }
    if (tid == 0) { unsigned long long* o = out + blockIdx.x * 4; o[0] = t_d; o[1] = t_a; o[2] = t_p; o[3] = t_l; }
    if (sink == 12345.678f) out[0] = 0;
}

int main() {
    float* buf; unsigned* sync; unsigned long long* out;
    const size_t bb = 2 * 2 * 128 * 64 * 4;
    const int iters = 3000, nwg = 2 * NG;
    hipMalloc(&buf, bb); hipMalloc(&sync, 2 * SYNC_STRIDE * 4); hipMalloc(&out, 256 * 4 * 8);
    hipFuncSetAttribute((const void*)xchg, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(buf, 0, bb); hipMemset(sync, 0, 2 * SYNC_STRIDE * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(xchg, dim3(nwg), dim3(256), 82 * 1024, 0, buf, sync, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwg * 4);
        hipMemcpy(h.data(), out, nwg * 4 * 8, hipMemcpyDeviceToHost);
        double s[4] = {0, 0, 0, 0}, sr[4] = {0, 0, 0, 0};
        int nr = 0;
        for (int i = 0; i < nwg; ++i) for (int k = 0; k < 4; ++k) { s[k] += h[i * 4 + k]; if (i % NG < NR) { sr[k] += h[i * 4 + k]; } }
        for (int i = 0; i < nwg; ++i) if (i % NG < NR) ++nr;
        if (rep) printf("NG=%d NR=%d LATE=%d RDELAY=%d WORK=%d  %6.3f us/iter  cycles (all WGs): drain %5.0f arrive %5.0f poll %5.0f load %5.0f | readers: poll %5.0f load %5.0f\n",
                        NG, NR, LATE, RDELAY, WORK, ms * 1e3 / iters, s[0] / nwg / iters, s[1] / nwg / iters, s[2] / nwg / iters, s[3] / nwg / iters,
                        sr[2] / nr / iters, sr[3] / nr / iters);
        fflush(stdout);
    }
    return 0;
}
