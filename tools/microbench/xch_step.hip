// What does ONE exchange step of the persistent recurrent kernels cost with no arithmetic at all?
// A group of G workgroups (256 threads) does what lstm_cluster / lstm_wide16 / mix_decoder do between two steps: it publishes
// its share of the step's h tile as tagged 16-byte granule pairs ({value, epoch, value, epoch}; the data is the flag), then
// gathers the partners' shares with sc1 (L1-bypassing) loads until every tag equals the epoch - and goes on to the next epoch.  Time per step = the floor of a recurrent step whose matrix work is free: the LATENCY FLOOR quoted by bench.py for the
// latency-bound configurations (configs[0], configs[4], lstm.py's shape).
//   mode same-XCD : the group's members sit on one XCD (blocks 8 apart), stores sc0 (stay in that XCD's L2)
//   mode safe     : members on neighbouring blocks = different XCDs, stores sc1 (write-through)
// build: hipcc --offload-arch=gfx950 -O3 -o build/xch_step tools/microbench/xch_step.hip ; run: build/xch_step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned SPIN_LIMIT = 1u << 22;

// The group's step tile is H units x 16 rows of 8-byte granules (H * 128 bytes); every workgroup publishes its 1/G of it in
// 16-byte pairs and gathers the other (G-1)/G, the 16-byte chunks dealt over its 256 lanes: NL loads per lane, ALL in flight
// before the first tag is looked at (as the kernels do), the whole set re-read while any tag is stale.
template <int NL>
__global__ __launch_bounds__(256) void xch_step_kernel(u32x4* buf, int steps, int G, int H, int same_xcd, unsigned* fail,
                                                       unsigned long long* cycles) {
    const int b = blockIdx.x, tid = threadIdx.x;
    int group, member;
    if (same_xcd) {          // block b runs on XCD b % 8: members of a group are 8 blocks apart
        const int xcd = b & 7, q = b >> 3;
        group = (q / G) * 8 + xcd;
        member = q % G;
    } else {
        group = b / G;
        member = b % G;
    }
    const int tile_chunks = H * 8;                    // 16-byte chunks of the tile
    const int own = tile_chunks / G;                  // chunks this workgroup publishes
    u32x4* gbase = buf + (size_t)group * 2 * tile_chunks;      // [parity][chunk]
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, 0x7fffffff, 0x00020000);
    unsigned long long t0 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memtime();
    bool dead = false;
    for (int e = 1; e <= steps && !dead; ++e) {
        const unsigned par = (unsigned)(e & 1) * (unsigned)tile_chunks;
        for (int k = tid; k < own; k += 256) {
            const u32x4 v = {(unsigned)(tid + k), (unsigned)e, (unsigned)(tid - k), (unsigned)e};
            const unsigned off = (par + (unsigned)(member * own + k)) * 16;
            if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 1 /* sc0 */);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16 /* sc1 */);
        }
        unsigned offs[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            int c = tid + 256 * l;                                     // chunk of the "others" region
            const bool ok = c < tile_chunks - own;
            c += (c >= member * own) ? own : 0;                        // skip this workgroup's own share
            offs[l] = ok ? (par + (unsigned)c) * 16 : (par + (unsigned)(member * own)) * 16;      // idle lanes re-read an own chunk (tag already e)
        }
        unsigned spins = 0;
        for (;;) {
            u32x4 v[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) v[l] = __builtin_amdgcn_raw_buffer_load_b128(rs, offs[l], 0, 16 /* sc1 */);
            bool ok = true;
#pragma unroll
            for (int l = 0; l < NL; ++l) ok = ok && v[l].y == (unsigned)e && v[l].w == (unsigned)e;
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { dead = true; break; }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        if (dead) break;
        __syncthreads();      // the step's barrier: the whole tile is in (as barrier 2 of the real kernels)
    }
    if (dead && tid == 0) atomicAdd(fail, 1u);
    if (tid == 0 && b == 0) cycles[0] = __builtin_amdgcn_s_memtime() - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NL>
int run(int groups, int G, int H, int same, int steps) {
    const int blocks = same ? ((groups + 7) / 8) * 8 * G : groups * G;      // same-XCD: whole rows of eight groups
    const int ngroups = same ? ((groups + 7) / 8) * 8 : groups;
    if ((H * 8 - H * 8 / G + 255) / 256 > NL) { printf("NL too small\n"); return 1; }
    u32x4* buf; unsigned* fail; unsigned long long* cyc;
    const size_t bytes = (size_t)ngroups * 2 * H * 8 * 16;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&cyc, 8));
    CK(hipMemset(fail, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {      // every launch restarts at epoch 1: zero the tags in between
        CK(hipMemset(buf, 0, bytes));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(xch_step_kernel<NL>, dim3(blocks), dim3(256), 0, 0, buf, steps, G, H, same, fail, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    unsigned f = 0; unsigned long long c = 0;
    CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-9s H=%3d G=%2d groups=%3d blocks=%4d  tile %2d KB, %4.1f KB published + %4.1f KB gathered per workgroup (%2d loads per lane): "
           "%6.3f us per step%s\n", same ? "same-XCD" : "safe", H, G, groups, blocks, H * 128 / 1024, H * 128.0 / G / 1024,
           H * 128.0 * (G - 1) / G / 1024, NL, (best * 1e3 - 6.0) / steps, f ? "  ** GAVE UP **" : "");
    CK(hipFree(buf)); CK(hipFree(fail)); CK(hipFree(cyc));
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    printf("one exchange step with no arithmetic: publish 1/G of a (16 rows x H units) tile of tagged 8-byte granules, gather the rest,\n"
           "one workgroup barrier; %d steps per launch, ~6 us of launch taken off; best of 3\n", steps);
    for (int same = 1; same >= 0; --same) {
        // H = 128: two workgroups (configs[0]); H = 256: four (headline kernel) or eight (wide layers, fused mixing decoder);
        // H = 512: sixteen / thirty-two (lstm.py's padded width)
        for (int groups : {2, 16}) if (run<2>(groups, 2, 128, same, steps)) return 1;
        for (int groups : {2, 8}) if (run<4>(groups, 8, 128, same, steps)) return 1;    // round 4: the reference-batch decode on eight workgroups per tile
        for (int groups : {1, 8, 64}) if (run<6>(groups, 4, 256, same, steps)) return 1;
        for (int groups : {1, 8, 32}) if (run<7>(groups, 8, 256, same, steps)) return 1;
        for (int groups : {1, 2, 8}) if (run<15>(groups, 16, 512, same, steps)) return 1;
        for (int groups : {1, 2, 8}) if (run<16>(groups, 32, 512, same, steps)) return 1;
    }
    return 0;
}
