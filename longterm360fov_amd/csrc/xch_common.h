// Workspace header and launch protocol shared by every kernel that exchanges {value, epoch} granules between
// workgroups (lstm_cluster, lstm_wide, lstm_bwd_cluster, mix_decoder, mix_decoder_bwd and their bf16 forms).
//
// Workspace layout (caller-owned, ZERO-FILLED ONCE after allocation - fov_workspace_init - never per call):
//     [0, 256)                 header words below
//     [256, 256 + kXchBytes)   granule / hello words; ONLY exchange kernels write here, always 8-byte words whose
//                              upper half is an epoch tag
//     [256 + kXchBytes, ...)   whatever else the entry point keeps in its workspace (packed weights, carried state)
//
// Epoch tags are MONOTONE ACROSS LAUNCHES: a launch reads the base from the header, every tag it writes lies in
// (base, base + span], and the last workgroup to ARRIVE (= the moment every workgroup has read the base) writes
// base + span for the next launch - no workgroup waits for another at either end of a kernel.  Stale granules of any earlier
// launch (of any kernel, any shape) therefore carry smaller tags than anything a later launch waits for, and the
// per-call memset of the granule area - a 5 us fill kernel in front of every launch - is gone.
//
// ST_TIMEOUT is sticky: a bounded wait that gives up sets it, every later launch on the workspace sees it on entry
// and skips its body (fail-stop: nothing is computed from half-exchanged state), and only fov_check_status clears it.
#pragma once
#include <hip/hip_runtime.h>

namespace fov {

enum : int {
    ST_TIMEOUT = 0,     // != 0: a bounded in-kernel wait gave up; cleared only by fov_check_status
    ST_ARRIVED = 1,     // workgroups of the running launch that have read the header (self-resetting)
    ST_LAUNCHES = 2,    // exchange launches whose workgroups have all arrived
    ST_EPOCH = 3,       // epoch base: every tag written by launches < ST_LAUNCHES is <= this value
    ST_SAFE0 = 4,       // ST_SAFE0 + (launch & 1): workgroups of that launch on the placement-independent (sc1) exchange
};

constexpr size_t kXchBytes = (size_t)64 << 20;   // fixed granule area: the largest user (fused decoder backward, 32 groups) needs 50.9 MB

__device__ __forceinline__ unsigned xch_status_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool xch_poisoned(const unsigned* status) { return xch_status_load(status + ST_TIMEOUT) != 0u; }
__device__ __forceinline__ void xch_give_up(unsigned* status) {
    __hip_atomic_store(status + ST_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What a workgroup takes from the header when it starts.  `base` and `launch` are uniform over the grid: the words are
// only rewritten by the LAST workgroup to arrive, i.e. after every workgroup of the launch has read them.
struct XchTicket {
    unsigned base;      // epoch base of this launch: its tags lie in (base, base + span]
    unsigned launch;    // index of this launch on the workspace
    unsigned arrival;   // thread 0 only: this workgroup's arrival number
};

// Kernel entry, in two halves around the prologue's first workgroup barrier: THREAD 0 reads the header, takes the
// arrival ticket and leaves base / launch in two LDS words (`lds2`); after the barrier every thread picks them up with
// xch_ticket().  Only one thread reads the header: a wave that starts late must not read it by itself - the last
// arriver (of another workgroup) may already have rewritten it for the next launch.
__device__ __forceinline__ unsigned xch_arrive(unsigned* status, unsigned* lds2) {
    unsigned arrival = 0;
    if (threadIdx.x == 0) {
        unsigned base = xch_status_load(status + ST_EPOCH);
        unsigned launch = xch_status_load(status + ST_LAUNCHES);
        // both words are read before the ticket is taken: the last arriver may rewrite them at once
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(base), "+v"(launch)::"memory");
        arrival = __hip_atomic_fetch_add(status + ST_ARRIVED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds2[0] = base;
        lds2[1] = launch;
    }
    return arrival;
}
__device__ __forceinline__ XchTicket xch_ticket(const unsigned* lds2, unsigned arrival) {
    // readfirstlane: the words are the same for every lane, but a value loaded from LDS is divergent to the compiler, and an
    // epoch-derived scalar offset of a buffer load / store (the parity buffer) then becomes a waterfall loop around EVERY
    // granule access (v_readfirstlane + v_cmp + s_and_saveexec + branch, 16 of them per step of lstm_cluster)
    XchTicket t;
#ifdef FOV_DBG_OLD_TICKET
    t.base = lds2[0];
    t.launch = lds2[1];
#else
    t.base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds2[0]);
    t.launch = (unsigned)__builtin_amdgcn_readfirstlane((int)lds2[1]);
#endif
    t.arrival = arrival;
    return t;
}
// Thread 0 calls it once, any time later (the kernels do at their very end, when the ticket's round trip is long over):
// the last arriver publishes the header of the NEXT launch.  Nothing here waits for the other workgroups.
__device__ __forceinline__ void xch_settle(unsigned* status, const XchTicket& t, unsigned span) {
    if (threadIdx.x == 0 && t.arrival == gridDim.x * gridDim.y * gridDim.z - 1u) {
        __hip_atomic_store(status + ST_ARRIVED, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(status + ST_SAFE0 + ((t.launch + 1u) & 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch's counter
        __hip_atomic_store(status + ST_EPOCH, t.base + span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(status + ST_LAUNCHES, t.launch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void xch_count_safe(unsigned* status, const XchTicket& t) {
    __hip_atomic_fetch_add(status + ST_SAFE0 + (t.launch & 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace fov
