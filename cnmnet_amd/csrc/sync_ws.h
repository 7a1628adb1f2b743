// Sync workspace of the persistent stream-K kernels (conv_winograd4s.hip, conv_rows_staged.hip): the cross-workgroup
// hand-off of partial outputs, and what happens when it does not complete.
//
// Layout (cnm_wino36_sync_floats() floats, zero before the FIRST use, one workspace per stream):
//   words 0 .. 1019   one flag per range (= workgroup) of a launch; zero between launches
//   bytes 4096 ..     one 128 KB partial-output slot per range
//
// A launch's GENERATION is its dispatch id -- the per-queue dispatch counter the command processor hands every wave in a user
// SGPR (llvm.amdgcn.dispatch.id): the same for all workgroups of a launch, different for every launch on the queue, eager or
// HIP-graph replay -- made odd so that it is never 0.  No memory traffic, no live register (it is re-read where needed).
// A range that publishes a partial output stores the generation in its flag; the range that finishes the unit polls for
// exactly that value and, having SEEN it, re-arms the flag to zero.  A consumer that gives up writes nothing, so a flag raised
// after its consumer timed out is merely a stale non-zero word carrying an old generation: it cannot satisfy a later launch's
// poll and is overwritten the next time that range publishes -- the workspace needs no repair (a flag lowered
// without having been seen raised used to poison every later launch on the stream).
// Two earlier forms of this round kept the generation in the workspace (advanced by the last workgroup out, found with a
// returning exit atomic; then: an exit count read at kernel start and parked in LDS).  Both were correct and both cost the bench
// step 0.7-1.3 % (tools/r4_sync_ab.sh, alternating builds on one box): 3-4 us of atomics per launch, and a dozen more
// spilled SGPRs that pushed the phase loop's spill lanes into a second VGPR.  This form compiles to the round-3 register
// allocation (tools/hotloop_proxy.sh: 63 spilled SGPRs, the same scratch reloads in the hot path).
//
// Failure is loud: a poll that exceeds its spin bound writes a non-zero word to a pinned host status word (system-scope
// store, no host synchronisation), and every staged-kernel entry point refuses to launch -- CNM_ERR_LAUNCH -- while that
// word is set; cnm_engine_status(1) reports and clears it.
#pragma once
#include "cnm_common.h"

constexpr size_t kSyncFlagBytes = 4096, kSyncSlotBytes = 8 * 16 * 64 * 16;
constexpr int kSyncMaxRanges = 1020;
constexpr unsigned kSyncFaultBit = 0x80000000u;                          // test-only fault injection: every wait fails as if its publisher had never come

struct SyncCtl {
    unsigned* status;                                                    // pinned host word (device-visible), or null
    unsigned spin_limit;                                                 // polls of a flag before giving up (| kSyncFaultBit); 0 = default
    unsigned version;                                                    // host bookkeeping: which upload this is
};
constexpr unsigned kSyncDefaultSpins = 1u << 24;                         // ~5 s of polling at s_sleep(8)

// Host side (defined in conv_winograd4s.hip).
SyncCtl cnm_sync_ctl(hipStream_t stream);                               // allocates the status word on first use outside a stream capture
bool cnm_sync_failed();                                                  // a timeout has been recorded and not yet cleared

#ifdef __HIPCC__
// The control block lives in device memory, one copy per translation unit and device, read on the cold paths only: as kernel
// arguments the three words stay live across the phase loop of kernels that have no register to spare (+20 spilled SGPRs).
static __device__ SyncCtl g_sync_ctl_dev;
// Host: make this translation unit's copy on the current device current (a blocking copy, the first time and after the test
// knob changed; skipped while `stream` is being captured -- a zero block means "default bound, no host word").
static inline void sync_ctl_upload(hipStream_t stream) {
    static unsigned uploaded[64] = {0};                                  // version + 1 per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    const SyncCtl c = cnm_sync_ctl(stream);
    if (uploaded[dev] == c.version + 1u) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sync_ctl_dev), &c, sizeof(c)) == hipSuccess) uploaded[dev] = c.version + 1u; else (void)hipGetLastError();
}

// The generation of this launch: the queue's dispatch id (a user SGPR the command processor fills in: the same for every
// workgroup of a launch, different for every launch on the queue, eager or graph replay), made odd so that it is never 0.
extern "C" __device__ unsigned long long cnm_llvm_dispatch_id() __asm("llvm.amdgcn.dispatch.id");   // no clang builtin for this intrinsic
__device__ static inline unsigned sync_generation() { return ((unsigned)cnm_llvm_dispatch_id() << 1) | 1u; }
// ONE lane: wait until range `idx` has published in this launch, then re-arm its flag.  A flag that was not seen is not written.
// The bound and the status pointer live in a device global and are looked at every 4096 polls only -- and once up front for the
// test-only fault bit, which makes the wait fail at once, as if its publisher had never come.
__device__ static inline void sync_give_up(int idx) {                    // loud: the result of this unit is wrong and the host is told
    unsigned* const status = *reinterpret_cast<unsigned* volatile*>(&g_sync_ctl_dev.status);
    if (status) __hip_atomic_store(status, 0x40000000u | (unsigned)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ static inline void sync_wait(unsigned* flags, int idx, unsigned gen) {
    unsigned spins = 0;
    for (;;) {
        if ((spins & 0xFFFu) == 0u) {                                    // first poll and every 4096th: the bound (and the test-only fault bit = bound 0)
            const unsigned ctl = *reinterpret_cast<volatile unsigned*>(&g_sync_ctl_dev.spin_limit);
            const unsigned limit = (ctl & kSyncFaultBit) ? 0u : (ctl ? ctl : kSyncDefaultSpins);
            if (spins >= limit) { sync_give_up(idx); return; }
        }
        if (__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) break;
        ++spins;
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(flags + idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static inline void sync_publish(unsigned* flags, int idx, unsigned gen) {   // ONE lane, after the workgroup's stores have drained
    __hip_atomic_store(flags + idx, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
