// Microbenchmark (diagnostic only): the h_t all-gather of the persistent scans, 2 groups x 128 workgroups,
// 32 KB per group and step ([128 blocks][16 b][4] floats, one 16-byte store per lane of wave 0's first 16 lanes...
// here: wave 0 lanes 0..15 store the workgroup's 256 B), every workgroup reads all 32 KB (8 b128 per lane, 4 waves).
//   V=0  counter protocol of scan_common.h (sc1 payload, drain, barrier, sharded+replicated counter, poll, sc1 loads)
//   V=1  sentinel protocol: 3 rotating slots, the producer resets its part of slot (s+2)%3 to a NaN pattern at the
//        top of step s, consumers re-read their share until no word equals the pattern; no drain, no counter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../asr_amd/csrc/scan_common.h"

#define NG 128
#define SENT 0xFFFFFFFFu
#ifndef CAN
#define CAN 7
#endif
#ifndef PSLEEP
#define PSLEEP 2
#endif
#ifndef WORK
#define WORK 0          // s_sleep units of fake compute between receive and publish
#endif
__device__ __forceinline__ float val(int it, int wg, int e) { return (float)(((it * 131 + wg * 7 + e) & 0xffff) + 1); }

template <int V>
__global__ __launch_bounds__(256) void xchg(float* buf, unsigned* sync, int iters, unsigned long long* out, unsigned* bad) {
    __shared__ int flag;
    const int wg = blockIdx.x % NG, grp = blockIdx.x / NG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int q = lane >> 4, bl = lane & 15;
    float* rex = buf + (size_t)grp * 3 * NG * 64;
    GroupSync gs;
    group_sync_init(gs, sync + grp * SYNC_STRIDE, NG, wg, &flag);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc_(rex, 3 * NG * 64 * 4);
    const unsigned ex_voff = (unsigned)((q * 64 + bl * 4) * 4);
    unsigned long long t_a = 0, t_b = 0;
    unsigned nbad = 0;
    // init: slot 0 = values of "step -1", slots 1,2 = sentinel (V=1); counter barrier once
    if (wave == 0 && q == 0) {
        f32x4 v0 = {val(-1, wg, bl * 4), val(-1, wg, bl * 4 + 1), val(-1, wg, bl * 4 + 2), val(-1, wg, bl * 4 + 3)};
        st_sc1_f4(rs, (unsigned)(((0 * NG + wg) * 64 + bl * 4) * 4), v0);
        const float sn = __uint_as_float(SENT);
        f32x4 s4 = {sn, sn, sn, sn};
        if (V != 0) {
            st_sc1_f4(rs, (unsigned)(((1 * NG + wg) * 64 + bl * 4) * 4), s4);
            st_sc1_f4(rs, (unsigned)(((2 * NG + wg) * 64 + bl * 4) * 4), s4);
        }
    }
    group_publish(gs);
    if (!group_wait(gs)) return;
    bool giveup = false;
    unsigned nretry = 0;
    for (int it = 0; it < iters && !giveup; ++it) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const int slot = V != 0 ? it % 3 : (it & 1);
        const int wslot = V != 0 ? (it + 1) % 3 : ((it + 1) & 1);
        f32x4 bv[8];
        if (V != 0 && wave == 0 && q == 0 && it > 0) {     // reset of the slot two steps ahead, early: everyone has
            const float sn = __uint_as_float(SENT);        // read it (we hold all of step it-1 ... see DESIGN)
        }
        if (V == 0) {
            if (it > 0) { if (!group_wait(gs)) break; }
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[i] = ldb_sc1_f4(rs, ex_voff, (unsigned)(slot * NG * 256) + (unsigned)(wv * 8 + i) * 1024u);
        } else {
            int spins = 0;
            if (V == 2) {                                  // canary: the wave's last block only, until it is fresh
                for (;;) {
                    asm volatile("" ::: "memory");
                    const f32x4 c = ldb_sc1_f4(rs, ex_voff, (unsigned)(slot * NG * 256) + (unsigned)(wv * 8 + (CAN)) * 1024u);
                    bool ok = true;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ok &= (__float_as_uint(c[e]) != SENT);
                    if (__all(ok)) break;
                    if (++spins > 20000) { giveup = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            for (;;) {
                asm volatile("" ::: "memory");            // the loads must be re-issued every pass (not hoisted)
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = ldb_sc1_f4(rs, ex_voff, (unsigned)(slot * NG * 256) + (unsigned)(wv * 8 + i) * 1024u);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ok &= (__float_as_uint(bv[i][e]) != SENT);
                if (__all(ok)) break;
                if (++spins > 20000) { nbad += 1000000; giveup = true; break; }
                if (V == 3) __builtin_amdgcn_s_sleep(PSLEEP);
                if (V == 2) nretry++;
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        // check: block (wv*8+i)*4+q is workgroup index; values of step it-1
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (bv[i][e] != val(it - 1, (wv * 8 + i) * 4 + q, bl * 4 + e)) nbad++;
        if (giveup) { if (lane == 0) out[512 + blockIdx.x * 4 + wave] = (unsigned long long)it; }
        if (V != 0 && wave == 0 && q == 0) {               // reset two slots ahead, issued right after the receive
            const float sn = __uint_as_float(SENT);
            st_sc1_f4(rs, (unsigned)(((((it + 2) % 3) * NG + wg) * 64 + bl * 4) * 4), f32x4{sn, sn, sn, sn});
        }
        giveup = __syncthreads_or(giveup);
        if (giveup) break;
        if (WORK) __builtin_amdgcn_s_sleep(WORK);
        if (wave == 0 && q == 0) {
            if (V != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // reset acknowledged before this step's data
            f32x4 v0 = {val(it, wg, bl * 4), val(it, wg, bl * 4 + 1), val(it, wg, bl * 4 + 2), val(it, wg, bl * 4 + 3)};
            st_sc1_f4(rs, (unsigned)(((wslot * NG + wg) * 64 + bl * 4) * 4), v0);
        }
        if (V == 0) group_publish(gs);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_a += t1 - t0; t_b += t2 - t1;
    }
    if (tid == 0) { out[blockIdx.x * 2 + 0] = t_a; out[blockIdx.x * 2 + 1] = t_b; }
    if (nbad) atomicAdd(bad, nbad);
    if (nretry) atomicAdd(bad + 1, nretry);
}

template <int V> void run(const char* name, int iters) {
    float* buf; unsigned* sync; unsigned long long* out; unsigned* bad;
    const size_t bb = 2 * 3 * NG * 64 * 4;
    hipMalloc(&buf, bb); hipMalloc(&sync, 2 * SYNC_STRIDE * 4); hipMalloc(&out, 256 * 6 * 8); hipMemset(out, 0xff, 256 * 6 * 8); hipMalloc(&bad, 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(buf, 0, bb); hipMemset(sync, 0, 2 * SYNC_STRIDE * 4); hipMemset(bad, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(xchg<V>, dim3(256), dim3(256), 82 * 1024, 0, buf, sync, iters, out, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 2); unsigned hb, hr;
        hipMemcpy(h.data(), out, 256 * 2 * 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&hr, bad + 1, 4, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < 256; ++i) { a += h[i * 2]; b += h[i * 2 + 1]; }
        if (hb) { std::vector<unsigned long long> g(256 * 6); hipMemcpy(g.data(), out, 256 * 6 * 8, hipMemcpyDeviceToHost);
            int shown = 0; for (int i = 0; i < 1024 && shown < 8; ++i) if (g[512 + i] != ~0ull) { printf("  gave up: wg %d wave %d at it %llu\n", i / 4, i % 4, g[512 + i]); ++shown; } }
        fflush(stdout);
        if (rep) printf("%-22s WORK=%d %7.3f us/iter   cycles/iter: receive %6.0f rest %6.0f   mismatches %u retries %u\n", name, WORK,
                        ms * 1e3 / iters, a / 256 / iters, b / 256 / iters, hb, hr);
    }
    hipFree(buf); hipFree(sync); hipFree(out); hipFree(bad);
}

int main() {
    hipFuncSetAttribute((const void*)xchg<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
    hipFuncSetAttribute((const void*)xchg<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
    hipFuncSetAttribute((const void*)xchg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
    hipFuncSetAttribute((const void*)xchg<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024);
    const int iters = 3000;
    run<0>("counter protocol", iters); fflush(stdout);
    run<1>("sentinel full poll", iters); fflush(stdout);
    run<2>("sentinel canary", iters); fflush(stdout);
    run<3>("sentinel poll+sleep", iters); fflush(stdout);
    return 0;
}
