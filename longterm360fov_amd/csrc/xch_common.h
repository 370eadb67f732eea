// Workspace header and launch protocol shared by every kernel that exchanges {value, epoch} granules between
// workgroups (lstm_cluster, lstm_wide, lstm_bwd_cluster, mix_decoder, mix_decoder_bwd and their bf16 forms).
//
// Workspace layout (caller-owned, ZERO-FILLED ONCE after allocation - fov_workspace_init - never per call):
//     [0, 256)                 header words below
//     [256, 256 + kXchBytes)   granule / hello words; ONLY exchange kernels write here, always 8-byte words whose
//                              upper half is an epoch tag (since the end of round 3 usually two of them per 16-byte
//                              store / load: neighbours in the kernel's granule order, each checked on its own tag)
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

// A granule store of an exchanging kernel: sc0 (the line stays in the XCD's L2, where the partners' L1-bypassing sc1 loads
// find it at L2-hit latency) when the hello handshake of this launch has shown every member of the group on ONE XCD
// (XchTicket::same_xcd), sc1 (write-through, placement-independent) otherwise.  Measured with the handshake forced off /
// on, same box (round 3): configs[2] training step 0.968 -> 0.884 ms fp32, 0.622 -> 0.545 ms bf16 operands.
#define XCH_STORE_B64(same, data, rsrc, voff, soff)                                                    \
    do {                                                                                               \
        if (same) __builtin_amdgcn_raw_buffer_store_b64(data, rsrc, voff, soff, 1 /* sc0 */);          \
        else __builtin_amdgcn_raw_buffer_store_b64(data, rsrc, voff, soff, 16 /* sc1 */);              \
    } while (0)

namespace fov {

enum : int {
    ST_TIMEOUT = 0,     // != 0: a bounded in-kernel wait gave up; cleared only by fov_check_status
                        // (word 32 was the arrival count up to round 4; the arrival words now sit behind the hello words)
    ST_LAUNCHES = 2,    // exchange launches whose workgroups have all arrived
    ST_EPOCH = 3,       // epoch base: every tag written by launches < ST_LAUNCHES is <= this value
    ST_SAFE0 = 48,      // ST_SAFE0 + (launch & 1): workgroups of that launch on the placement-independent (sc1) exchange (byte 192: third line)
    ST_FORCE_SAFE = 6,  // != 0: never take the same-XCD fast exchange (fov_workspace_force_safe: the tests' A/B switch)
};

constexpr size_t kXchBytes = (size_t)64 << 20;   // fixed granule area: the largest user (fused decoder backward, 32 groups) needs 50.9 MB
// Arrival words: workgroups of the running launch that have read the header (reset by xch_settle), dealt over kArriveWords
// words 128 bytes apart by linear block index.  ONE word (up to round 4) cost every launch microseconds: the add is
// fire-and-forget, but it stays in the wave's vmcnt queue until the memory side acknowledges it, vmcnt retires in order, and
// N adds on one word drain one after the other - the first load a workgroup waits for behind its add (its weights) was held
// 1.8 us (median) to 3.0 us (last workgroup) in a 256-workgroup launch, 3.7 / 6.3 us in the 512-workgroup two-layer kernel,
// against 0.12 us with no add in front; with 64 words 0.46 / 0.59 us (tools/microbench/arrive_fanin.hip, round 4).
constexpr int kArriveWords = 64;
constexpr int kArriveStride = 32;                                      // in 32-bit words
constexpr size_t kArriveBytes = (size_t)kArriveWords * kArriveStride * 4;   // 8 KB
constexpr size_t kHelloBytes = ((size_t)64 << 10) + kArriveBytes;      // the area's tail: 64 KB of hello words of the same-XCD handshake,
                                                                       // [group][32 members], then the arrival words
constexpr int kHelloStride = 32;                 // members per group at most (width-512 layer: 16 or 32 workgroups per tile); 256 groups fit

// where the members of group `group` publish {launch tag, XCC id}: only the status pointer (= workspace start) is needed
__device__ __forceinline__ unsigned long long* xch_hello_words(unsigned* status, int group) {
    return (unsigned long long*)((char*)status + 256 + kXchBytes - kHelloBytes) + (size_t)group * kHelloStride;
}
__device__ __forceinline__ unsigned* xch_arrive_word(unsigned* status, unsigned i) {
    return (unsigned*)((char*)status + 256 + kXchBytes - kArriveBytes) + (size_t)i * kArriveStride;
}
__device__ __forceinline__ unsigned xch_linear_block() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ unsigned xch_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}

__device__ __forceinline__ unsigned xch_status_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool xch_poisoned(const unsigned* status) { return xch_status_load(status + ST_TIMEOUT) != 0u; }
// the timeout word as loaded, and the test of it at the place of use: `xch_poisoned()` at a kernel's entry is hoisted, compare
// and wait included, in front of everything the kernel requests next
__device__ __forceinline__ unsigned xch_timeout_word(const unsigned* status) { return xch_status_load(status + ST_TIMEOUT); }
__device__ __forceinline__ bool xch_timeout_set(unsigned word) {
    asm volatile("" : "+v"(word)::"memory");
    return word != 0u;
}
__device__ __forceinline__ void xch_give_up(unsigned* status) {
    __hip_atomic_store(status + ST_TIMEOUT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What a workgroup takes from the header when it starts.  `base` and `launch` are uniform over the grid: the words are
// only rewritten after every workgroup of the launch has read them (xch_settle).
struct XchTicket {
    unsigned base;      // epoch base of this launch: its tags lie in (base, base + span]
    unsigned launch;    // index of this launch on the workspace
    unsigned arrival;   // unused since round 3 (kept for the call sites' signature)
    bool same_xcd;      // the hello handshake of this launch found every member of the group on one XCD (XCH_STORE_B64)
};

// Kernel entry, in two halves around the prologue's first workgroup barrier: THREAD 0 reads the header, counts its
// workgroup as arrived and leaves base / launch in two LDS words (`lds2`); after the barrier every thread picks them up
// with xch_ticket().  Only one thread reads the header: a wave that starts late must not read it by itself - it may
// already have been rewritten for the next launch.
// The arrival is a fire-and-forget add (round 3).  It used to be a RETURNING add - "the last arriver settles" - which every
// workgroup waited for at its entry: 256 returning adds on one word take 3 us to drain (11-13 ns each,
// MI355X_MICROARCH.md price list, row fanin), in front of every exchange launch.  Round 4: even unreturned it was waited
// for, by the in-order vmcnt of the loads behind it - hence one of kArriveWords words (comment at kArriveWords).
__device__ __forceinline__ void xch_count_arrival(unsigned* status) {
    __hip_atomic_fetch_add(xch_arrive_word(status, xch_linear_block() % kArriveWords), 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);   // result unused: no return
}
// `lds2`: FOUR words since round 3 - base, launch, "hello handshake in use", "a partner sits on another XCD / fast path
// forbidden".  group >= 0 starts the same-XCD handshake of the group's members (blocks 8 apart under the kernels' block
// -> (group, slice) map, which round-robin dispatch puts on one XCD - a placement PREFERENCE that the handshake verifies
// at run time): thread 0 publishes {launch tag, HW_REG_XCC_ID} at once, xch_hello_poll() collects the partners' words
// later, behind the weight loads, when they have long become visible.
__device__ __forceinline__ unsigned xch_arrive(unsigned* status, unsigned* lds2, int group = -1, int slice = 0) {
    if (threadIdx.x == 0) {
        unsigned base = xch_status_load(status + ST_EPOCH);
        unsigned launch = xch_status_load(status + ST_LAUNCHES);
        unsigned force = xch_status_load(status + ST_FORCE_SAFE);
        // the words are read before the workgroup counts as arrived: once all have, the header may be rewritten
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(base), "+v"(launch), "+v"(force)::"memory");
        xch_count_arrival(status);
        lds2[0] = base;
        lds2[1] = launch;
        lds2[2] = group >= 0 ? 1u : 0u;
        lds2[3] = force;
        if (group >= 0)   // the tag is larger than any tag of an earlier launch (every launch advances the base by >= 2)
            __hip_atomic_store(xch_hello_words(status, group) + slice, ((unsigned long long)(base + 1u) << 32) | xch_xcc_id(),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return 0u;
}
// xch_arrive in two halves (round 4) for a kernel that loads weights in its prologue: thread 0 REQUESTS the header words at
// the kernel's entry, the first batch of weight loads goes out behind them, and only then does it take the words, count the
// workgroup as arrived and publish its hello word - the header's round trip (0.25 us) runs under the weight loads, and the
// add's and the store's acknowledgements (0.45 us, waited for by whatever load comes next: vmcnt retires in order) under
// the weight batches that follow.  In one piece the two sat in front of the first weight load: 2 500-3 400 cycles from the
// kernel's entry to its first weight request (tools/stamp_bf16_layer.py --stack2).
// Both halves are BRANCH-FREE for the vector memory pipe: EVERY thread executes the loads, the add and the store, all
// lanes but thread 0 with an offset beyond the descriptor (loads return 0, the add and the store are dropped by the bounds
// check).  Inside `if (threadIdx.x == 0)` the waves that skip the branch and the one that takes it have different numbers
// of operations in flight at the merge, and the compiler then waits for the weight batch behind it with vmcnt(0) - add's
// acknowledgement included.
struct XchHeader {
    unsigned base, launch, force;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xch_status_rsrc(unsigned* status) {
    return __builtin_amdgcn_make_buffer_rsrc(status, 0, (int)(256 + kXchBytes), 0x00020000);
}
// active == false (a launch that exchanges nothing): every lane is out of range, nothing is read, counted or published
__device__ __forceinline__ XchHeader xch_arrive_request(unsigned* status, bool active = true) {
    const __amdgpu_buffer_rsrc_t rs = xch_status_rsrc(status);
    const unsigned off = (threadIdx.x == 0 && active) ? 0u : 0x80000000u;
    XchHeader h;
    h.base = __builtin_amdgcn_raw_buffer_load_b32(rs, off + ST_EPOCH * 4u, 0, 16 /* sc1 */);
    h.launch = __builtin_amdgcn_raw_buffer_load_b32(rs, off + ST_LAUNCHES * 4u, 0, 16);
    h.force = __builtin_amdgcn_raw_buffer_load_b32(rs, off + ST_FORCE_SAFE * 4u, 0, 16);
    return h;
}
__device__ __forceinline__ void xch_arrive_commit(unsigned* status, unsigned* lds2, const XchHeader& h, int group = -1, int slice = 0,
                                                  bool active = true) {
    const __amdgpu_buffer_rsrc_t rs = xch_status_rsrc(status);
    const unsigned off = (threadIdx.x == 0 && active) ? 0u : 0x80000000u;
    // the words are read before the workgroup counts as arrived (once all have, the header may be rewritten): the add's
    // operand is made to depend on them, so the wait for the three loads - and for nothing younger - precedes it
    // (the words are also opaque to the compiler up to here: it hoisted `base + 1` and with it the wait for the loads to
    // the kernel's first block, in front of every weight request)
    int one = 1;
    unsigned base = h.base, launch = h.launch, force = h.force;
    asm volatile("" : "+v"(one), "+v"(base), "+v"(launch), "+v"(force)::"memory");
    const unsigned arrive_off = (unsigned)(256 + kXchBytes - kArriveBytes) + (xch_linear_block() % kArriveWords) * (unsigned)(kArriveStride * 4);
    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(one, rs, off + arrive_off, 0, 0);   // agent scope, no return (as xch_count_arrival)
    const unsigned hello_off = (unsigned)(256 + kXchBytes - kHelloBytes) + (unsigned)((group < 0 ? 0 : group) * kHelloStride + slice) * 8u;
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64((u32x2_){xch_xcc_id(), base + 1u}, rs, (group >= 0 ? off : 0x80000000u) + hello_off, 0, 16 /* sc1 */);
    if (threadIdx.x == 0 && active) {   // (LDS only: the vector memory counters are the same on both sides of this branch)
        lds2[0] = base;
        lds2[1] = launch;
        lds2[2] = group >= 0 ? 1u : 0u;
        lds2[3] = force;
    }
}
// Every thread may call it (in front of the barrier that precedes xch_ticket); lanes 0..G-1 of wave 0 each wait, bounded,
// for one member's hello word of THIS launch and record a foreign XCC id.  A give-up poisons the workspace and sets *abort.
__device__ __forceinline__ void xch_hello_poll(unsigned* status, unsigned* lds2, int group, int G, int* abort_flag) {
    if (threadIdx.x < (unsigned)G && lds2[2] != 0u) {   // (same wave as thread 0, whose LDS stores precede in program order)
        const unsigned long long* hello = xch_hello_words(status, group) + threadIdx.x;
        const unsigned tag = lds2[0] + 1u;
        unsigned long long hv = 0;
        unsigned spins = 0;
        while (true) {
            hv = __hip_atomic_load(hello, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(hv >> 32) == tag) break;
            if (++spins > (1u << 20) || ((spins & 63u) == 0 && xch_poisoned(status))) {
                xch_give_up(status);
                *abort_flag = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if ((unsigned)hv != xch_xcc_id()) lds2[3] = 1u;
    }
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
    t.same_xcd = __builtin_amdgcn_readfirstlane((int)lds2[2]) != 0 && __builtin_amdgcn_readfirstlane((int)lds2[3]) == 0;
    return t;
}
// Every workgroup calls it at its very end; the first wave of BLOCK 0 publishes the header of the NEXT launch once the
// arrival words show that every workgroup of this launch has read the current one - hundreds of microseconds ago for a
// persistent kernel, so the one load a lane takes is the whole cost (lane i watches word i, which counts the workgroups
// whose linear index is i modulo kArriveWords).  The wait is bounded like every other one: a grid that is not co-resident
// (block 0 done before the last block could start) poisons the workspace instead of hanging.
__device__ __forceinline__ void xch_settle(unsigned* status, const XchTicket& t, unsigned span) {
    static_assert(kArriveWords == 64, "one lane of block 0's first wave per arrival word");
    if (threadIdx.x < (unsigned)kArriveWords && xch_linear_block() == 0) {
        const unsigned grid = gridDim.x * gridDim.y * gridDim.z;
        const unsigned expect = grid / kArriveWords + (threadIdx.x < grid % kArriveWords ? 1u : 0u);
        unsigned* word = xch_arrive_word(status, threadIdx.x);
        unsigned spins = 0;
        bool ok = true;
        while (xch_status_load(word) != expect) {
            if (++spins > (1u << 20)) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (__any(!ok)) {
            if (threadIdx.x == 0) xch_give_up(status);
            return;
        }
        __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            __hip_atomic_store(status + ST_SAFE0 + ((t.launch + 1u) & 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch's counter
            __hip_atomic_store(status + ST_EPOCH, t.base + span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status + ST_LAUNCHES, t.launch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// Grids of G-member groups are padded to a multiple of eight groups: block b is member (b / 8) % G of group
// (b / (8 G)) * 8 + b % 8, so a group's members sit 8 blocks apart - one XCD under round-robin dispatch - whatever the group
// count (round 4, late: a count that is no multiple of 8 used to deal every group over all XCDs; an exchange step costs 1.4 us
// more there).  A block of an absent group counts as arrived (xch_settle expects the whole grid) and leaves.
__host__ __device__ constexpr int xch_padded_groups(int num_groups) { return (num_groups + 7) & ~7; }
__device__ __forceinline__ void xch_spare_leaves(unsigned* status, bool exchanging = true) {
    __shared__ unsigned sSpare[4];
    if (exchanging) xch_arrive(status, sSpare, -1, 0);
}
__device__ __forceinline__ void xch_count_safe(unsigned* status, const XchTicket& t) {
    __hip_atomic_fetch_add(status + ST_SAFE0 + (t.launch & 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace fov
