// Sync workspace of the persistent stream-K kernels (conv_winograd4s.hip, conv_rows_staged.hip): the cross-workgroup
// hand-off of partial outputs, and what happens when it does not complete.
//
// Layout (cnm_wino36_sync_floats() floats, zero before the FIRST use, one workspace per stream):
//   words 0 .. 1019   one flag per range (= workgroup) of a launch; zero between launches
//   word  1020        exit count: every workgroup adds 1 when it leaves (never reset: it only grows)
//   bytes 4096 ..     one 128 KB partial-output slot per range
//
// A launch's GENERATION is the exit count its workgroups read when they start (+1): launches on a stream are ordered, so
// every workgroup of a launch reads the same value, and every launch -- eager or a HIP-graph replay -- reads a different
// one.  A range that publishes a partial output stores the generation in its flag; the range that finishes the unit polls
// for exactly that value and, having SEEN it, re-arms the flag to zero.  A consumer that gives up writes nothing, so a
// flag raised after its consumer timed out is merely a stale non-zero word: it carries an old generation, can never
// satisfy a later launch's poll, and is overwritten the next time that range publishes -- the workspace needs no repair
// (a flag lowered without having been seen raised used to poison every later launch on the stream).
// Cost against the 0 / 1 flags of round 3: one load at kernel start (consumed much later) and one fire-and-forget
// atomic at exit.  (A first version advanced the generation with a returning exit atomic + last-leaver logic: 3-4 us per
// launch, 1.3 % of the bench step -- tools/r4_sync_ab.sh.)  If workgroups of one launch are so far apart that one leaves
// before another starts, the late one reads a different generation and the hand-off times out: a loud failure, not a
// wrong result.
//
// Failure is loud: a poll that exceeds its spin bound writes a non-zero word to a pinned host status word (system-scope
// store, no host synchronisation), and every staged-kernel entry point refuses to launch -- CNM_ERR_LAUNCH -- while that
// word is set; cnm_engine_status(1) reports and clears it.
#pragma once
#include "cnm_common.h"

constexpr size_t kSyncFlagBytes = 4096, kSyncSlotBytes = 8 * 16 * 64 * 16;
constexpr int kSyncMaxRanges = 1020, kSyncExitWord = 1020;
constexpr unsigned kSyncFaultBit = 0x80000000u;                          // test-only fault injection: publishers keep their flag down

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

// ONE lane, at kernel start: the generation of this launch (never 0).  The caller parks it in LDS; it is needed at unit ends only.
__device__ static inline unsigned sync_generation(const unsigned* flags) {
    const unsigned g = __hip_atomic_load(flags + kSyncExitWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    return g ? g : 1u;
}
// ONE lane: wait until range `idx` has published in this launch, then re-arm its flag.  A flag that was not seen is not written.
__device__ static inline void sync_wait(unsigned* flags, int idx, unsigned gen) {
    unsigned* const status = *reinterpret_cast<unsigned* volatile*>(&g_sync_ctl_dev.status);
    unsigned limit = *reinterpret_cast<volatile unsigned*>(&g_sync_ctl_dev.spin_limit) & ~kSyncFaultBit;
    if (!limit) limit = kSyncDefaultSpins;
    unsigned spins = 0;
    while (__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        if (++spins > limit) {                                           // give up loudly: the result of this unit is wrong and the host is told
            if (status) __hip_atomic_store(status, 0x40000000u | (unsigned)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(flags + idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static inline void sync_publish(unsigned* flags, int idx, unsigned gen) {   // ONE lane, after the workgroup's stores have drained
    if (!(*reinterpret_cast<volatile unsigned*>(&g_sync_ctl_dev.spin_limit) & kSyncFaultBit)) __hip_atomic_store(flags + idx, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ONE lane per workgroup, as its last action: fire and forget (the result is not used, nothing waits for it)
__device__ static inline void sync_leave(unsigned* flags) {
    (void)__hip_atomic_fetch_add(flags + kSyncExitWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
