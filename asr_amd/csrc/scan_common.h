// Device-side primitives shared by the persistent recurrent-scan kernels.
//
// A "group" = the G workgroups that cooperate on one layer-direction (x one chunk of <= 16 batch
// rows).  Each workgroup owns UT tiles of 4 units (= 16 gate columns in the engine's gate-
// interleaved column order  col = unit*4 + gate) and keeps its slice of the recurrent matrix in LDS
// for the whole sequence.  Between dependent phases the group exchanges small vectors through
// global memory with the placement-independent protocol of cdna_hip_programming.md Guideline 16:
//   producer : payload stored write-through (sc1), every storing wave drains vmcnt(0), workgroup
//              barrier, ONE lane adds 1 to the group's monotonic arrival counter (agent scope);
//   consumer : ONE lane polls the counter with sc1 loads until it reaches epoch*G, workgroup barrier,
//              then EVERY load of the handed-off bytes is an sc1 load (no L1 involvement).
// Every wait is bounded by its own wall-clock deadline (10 s from its first unsuccessful poll); on expiry the group's abort word is raised, every
// workgroup leaves its time loop, and the host reports ASR_ERR_TIMEOUT from asr_scan_status().
#pragma once
#include "common.h"
#include <stdlib.h>

typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

#define SYNC_SHARDS 16
#define SYNC_REPL 8                            // replicas of the sharded counter: a workgroup ADDS to its
                                               // shard of every replica (one instruction, 8 lanes) and
                                               // POLLS only replica wg % 8, so a line has G/16 adders and
                                               // G/8 pollers instead of G pollers (MI355X_MICROARCH.md,
                                               // polling-cost: 255 pollers of one word = 4 us)
#define SYNC_STRIDE ((SYNC_REPL * SYNC_SHARDS + 1) * 32)   // u32 words per group, every counter and the
                                                           // abort word on a 128-B line of its own
#define SYNC_ABORT (SYNC_REPL * SYNC_SHARDS * 32)
#define SCAN_DEADLINE_TICKS 1000000000ull      // 10 s of s_memrealtime (100 MHz)

// One arrival counter serialises G atomics at the memory side (~12 ns each: measured 2.1 us of a
// 3.6 us exchange at G = 128, tools/ubench/xchg.hip); 16 shards cut the fan-in to G/16 per word.
struct GroupSync {
    unsigned* words;            // this group's sync block
    unsigned G;                 // workgroups in the group
    unsigned epoch;             // exchanges completed so far (uniform across the group)
    unsigned my_shard;          // shard this workgroup arrives on
    unsigned my_repl;           // replica this workgroup polls
    unsigned members;           // lane s < SYNC_SHARDS: number of workgroups arriving on shard s
    unsigned long long deadline;
    int* lds_flag;              // one LDS int for broadcasting the poll result
};

__device__ __forceinline__ void group_sync_init(GroupSync& gs, unsigned* words, unsigned G, unsigned wg,
                                                int* lds_flag) {
    gs.words = words; gs.G = G; gs.epoch = 0; gs.lds_flag = lds_flag;
    gs.my_shard = wg % SYNC_SHARDS;
    gs.my_repl = wg % SYNC_REPL;
    const unsigned lane = threadIdx.x & 63;
    gs.members = lane < SYNC_SHARDS ? (G > lane ? (G - lane + SYNC_SHARDS - 1) / SYNC_SHARDS : 0) : 0;
    gs.deadline = __builtin_amdgcn_s_memrealtime() + SCAN_DEADLINE_TICKS;
}

__device__ __forceinline__ void group_publish(GroupSync& gs) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its sc1 stores
    __syncthreads();
    if (threadIdx.x < SYNC_REPL)
        __hip_atomic_fetch_add(gs.words + (threadIdx.x * SYNC_SHARDS + gs.my_shard) * 32, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// The same arrival when only ONE wave of the workgroup stored payload for this exchange: that wave
// drains and arrives by itself, the other waves neither wait nor meet at a barrier (their loads stay
// in flight).  `storing_wave` is wave-uniform.
__device__ __forceinline__ void group_publish_wave(GroupSync& gs, bool storing_wave) {
    if (storing_wave) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) < SYNC_REPL)
            __hip_atomic_fetch_add(gs.words + ((threadIdx.x & 63) * SYNC_SHARDS + gs.my_shard) * 32, 1u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// returns false when the group aborted (timeout somewhere): caller must leave its loop.  The deadline bounds
// THIS wait (armed at its first unsuccessful poll), not the launch: a long sequence, a profiler-serialised run
// or time slicing with another process cannot trip it.
__device__ __forceinline__ bool group_wait(GroupSync& gs) {
    gs.epoch += 1;
    if ((threadIdx.x >> 6) == (blockDim.x >> 6) - 1) {    // the LAST wave polls (it owns no per-step global
        const unsigned lane = threadIdx.x & 63;           // loads that would queue in front of the poll):
                                                          // lane s watches shard s, lane 16 the abort word
        const unsigned target = gs.epoch * gs.members;
        int good = 1;
        bool armed = false;
        for (;;) {
            unsigned v = 0;
            if (lane <= SYNC_SHARDS)
                v = ld_sc1_u32(lane < SYNC_SHARDS ? gs.words + (gs.my_repl * SYNC_SHARDS + lane) * 32 : gs.words + SYNC_ABORT);
            const bool arrived = lane < SYNC_SHARDS ? v >= target : true;
            const bool aborted = (lane == SYNC_SHARDS) && v != 0;
            if (__any(aborted)) { good = 0; break; }
            if (__all(arrived)) break;
            if (!armed) { gs.deadline = __builtin_amdgcn_s_memrealtime() + SCAN_DEADLINE_TICKS; armed = true; }
            else if (__builtin_amdgcn_s_memrealtime() > gs.deadline) {
                if (lane == 0)
                    __hip_atomic_store(gs.words + SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *gs.lds_flag = good;
    }
    __syncthreads();
    const int ok = *gs.lds_flag;
    __syncthreads();            // flag may be rewritten by the next wait
    return ok != 0;
}

// 16-byte sc1 load / store of exchanged payload (buffer_load/store_dwordx4 ... sc1)
__device__ __forceinline__ f32x4 ld_sc1_f4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void st_sc1_f4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 f) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), rs, byte_off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
// 4-byte sc1 load with a per-lane byte offset (VGPR, loop-invariant) and a wave-uniform byte offset
// (SGPR): no per-load vector address arithmetic; out-of-range reads return 0 (buffer range check).
__device__ __forceinline__ float ldb_sc1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 16));
}
__device__ __forceinline__ f32x4 ldb_sc1_f4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16));
}
// two floats as ONE aligned 8-byte sc1 store / load
__device__ __forceinline__ void st_sc1_f2(float* p, float a, float b) {
    unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ld_sc1_f2(const float* p, float& a, float& b) {
    unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((unsigned)v);
    b = __uint_as_float((unsigned)(v >> 32));
}

// sum over the 4 lanes that share a batch row inside one 16x16 tile (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float sum_q(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- host side: can a persistent grid of `nwg` workgroups with `lds` bytes of LDS each be co-resident?
// (fewer CUs than the full chip -- a compute partition, a CU mask, another process's persistent grid -- would
// leave workgroups un-dispatched and every group wait would run into its deadline: refuse up front instead)
static inline bool scan_grid_fits(size_t nwg, size_t lds) {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return false;
        cus[dev] = n;
    }
    const size_t per_cu = lds > 0 ? (160 * 1024) / lds : 1;
    return nwg <= (size_t)cus[dev] * (per_cu < 1 ? 1 : per_cu);
}

// ---- how many 16-row batch chunks one workgroup of a persistent scan may carry (teams of 4 waves on the same
// resident weights, one publish / wait per exchange for all of them): 4 unless ASR_SCAN_NC (read once) or
// asr_scan_set_chunks_per_wg (tests: the per-row results must not depend on it) says less
extern int asr_scan_nc_cap_;
static inline int scan_nc_cap() {
    static const int env_nc = [] { const char* e = getenv("ASR_SCAN_NC"); const int v = e ? atoi(e) : 4;
                                   return v >= 4 ? 4 : v >= 2 ? 2 : 1; }();
    return asr_scan_nc_cap_ ? asr_scan_nc_cap_ : env_nc;
}

// ---- optional in-kernel phase stamps (diagnostic build only: -DSCAN_STAMPS; never in the product .so)
#define STAMP_BYTES 65536
#ifdef SCAN_STAMPS
#define STAMP_DECL unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_prev; st_prev = n_; } while (0)
#define STAMP_DUMP(ptr, wg) do { if (threadIdx.x == 0 && (ptr) && (wg) < 512) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) (ptr)[(size_t)(wg) * 16 + i_] = st_acc[i_]; } } while (0)
#else
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#define STAMP_DUMP(ptr, wg) do { } while (0)
#endif
