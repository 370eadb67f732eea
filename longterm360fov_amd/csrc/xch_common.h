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
// (base, base + span], and the last workgroup to leave adds `span` to the base.  Stale granules of any earlier
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
    ST_SAFE_COUNT = 1,  // workgroups of the running launch that use the placement-independent (sc1) exchange
    ST_DONE = 2,        // workgroups of the running launch that have left
    ST_SAFE_LAST = 3,   // ST_SAFE_COUNT of the last completed launch (fov_exchange_mode)
    ST_EPOCH = 4,       // epoch base: every tag written so far is <= this value
    ST_LAUNCHES = 5,    // completed exchange launches (diagnostic)
};

constexpr size_t kXchBytes = (size_t)64 << 20;   // fixed granule area: the largest user (fused decoder backward, 32 groups) needs 50.9 MB

__device__ __forceinline__ unsigned xch_status_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// epoch base of this launch.  Uniform over the grid: the word is only rewritten by the last workgroup to leave.
__device__ __forceinline__ unsigned xch_epoch_base(const unsigned* status) { return xch_status_load(status + ST_EPOCH); }
__device__ __forceinline__ bool xch_poisoned(const unsigned* status) { return xch_status_load(status + ST_TIMEOUT) != 0u; }
__device__ __forceinline__ void xch_give_up(unsigned* status) {
    __hip_atomic_store(status + ST_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Every workgroup of an exchanging launch calls this exactly once, on every path, as its last action.
__device__ __forceinline__ void xch_leave(unsigned* status, unsigned span) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(status + ST_DONE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x * gridDim.y * gridDim.z - 1u) {
            const unsigned safe = __hip_atomic_exchange(status + ST_SAFE_COUNT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status + ST_SAFE_LAST, safe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status + ST_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(status + ST_LAUNCHES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(status + ST_EPOCH, span, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace fov
