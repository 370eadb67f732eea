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
// (base, base + span], and block 0 writes base + span for the next launch at its end, once the arrival count shows that
// every workgroup has read the base (a fire-and-forget add at entry) - no workgroup waits for another at the start of a
// kernel.  Stale granules of any earlier
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
    ST_ARRIVED = 1,     // workgroups of the running launch that have read the header (reset by xch_settle)
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
// only rewritten after every workgroup of the launch has read them (xch_settle).
struct XchTicket {
    unsigned base;      // epoch base of this launch: its tags lie in (base, base + span]
    unsigned launch;    // index of this launch on the workspace
    unsigned arrival;   // unused since round 3 (kept for the call sites' signature)
};

// Kernel entry, in two halves around the prologue's first workgroup barrier: THREAD 0 reads the header, counts its
// workgroup as arrived and leaves base / launch in two LDS words (`lds2`); after the barrier every thread picks them up
// with xch_ticket().  Only one thread reads the header: a wave that starts late must not read it by itself - it may
// already have been rewritten for the next launch.
// The arrival is a fire-and-forget add (round 3).  It used to be a RETURNING add - "the last arriver settles" - which every
// workgroup waited for at its entry: 256 returning adds on one word take 3 us to drain (11-13 ns each,
// MI355X_MICROARCH.md price list, row fanin), in front of every exchange launch.
__device__ __forceinline__ void xch_count_arrival(unsigned* status) {
    __hip_atomic_fetch_add(status + ST_ARRIVED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no return, no wait
}
__device__ __forceinline__ unsigned xch_arrive(unsigned* status, unsigned* lds2) {
    if (threadIdx.x == 0) {
        unsigned base = xch_status_load(status + ST_EPOCH);
        unsigned launch = xch_status_load(status + ST_LAUNCHES);
        // both words are read before the workgroup counts as arrived: once all have, the header may be rewritten
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(base), "+v"(launch)::"memory");
        xch_count_arrival(status);
        lds2[0] = base;
        lds2[1] = launch;
    }
    return 0u;
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
// Every workgroup calls it at its very end; thread 0 of BLOCK 0 publishes the header of the NEXT launch once the
// arrival count shows that every workgroup of this launch has read the current one - hundreds of microseconds ago for a
// persistent kernel, so the one load it takes is the whole cost.  The wait is bounded like every other one: a grid that
// is not co-resident (block 0 done before the last block could start) poisons the workspace instead of hanging.
__device__ __forceinline__ void xch_settle(unsigned* status, const XchTicket& t, unsigned span) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        const unsigned grid = gridDim.x * gridDim.y * gridDim.z;
        unsigned spins = 0;
        while (xch_status_load(status + ST_ARRIVED) != grid) {
            if (++spins > (1u << 20)) {
                xch_give_up(status);
                return;
            }
            __builtin_amdgcn_s_sleep(8);
        }
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
