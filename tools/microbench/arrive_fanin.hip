// What does the arrival count of an exchange launch cost a workgroup that does NOT wait for it?
// Every workgroup of a persistent exchange kernel adds 1 to an arrival word at its entry (xch_common.h, xch_count_arrival:
// fire-and-forget, no return value) and then goes on to load its weights.  The add stays in the wave's vmcnt queue until the
// memory side has acknowledged it, and vmcnt retires IN ORDER: the first s_waitcnt on a YOUNGER load also waits for the add.
// N adds on one word drain one after the other (MI355X_MICROARCH.md price list, row "fanin": 11-13 ns each), so the last
// workgroup's first load is held for N x 12 ns.  This measures, per workgroup, entry -> "a load issued behind the add has
// returned", with the adds dealt over W words `stride` bytes apart (W = 1: what the kernels did up to round 4).
// build: hipcc --offload-arch=gfx950 -O3 -o build/arrive_fanin tools/microbench/arrive_fanin.hip ; run: build/arrive_fanin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256) void arrive_kernel(unsigned* words, int W, int stride_words, const unsigned* other, int do_add,
                                                     unsigned long long* cycles, unsigned* sink) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned hdr = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the header read in front of the add
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(hdr)::"memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (do_add) __hip_atomic_fetch_add(words + (size_t)(blockIdx.x % W) * stride_words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned v = other[64 + blockIdx.x];                                                   // "the weights": a load behind the add
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        cycles[2 * blockIdx.x] = t1 - t0;
        cycles[2 * blockIdx.x + 1] = t2 - t1;
        sink[blockIdx.x] = v + hdr;
    }
}

int main() {
    unsigned *words, *other, *sink;
    unsigned long long* cycles;
    const int NB = 512;
    hipMalloc(&words, 1 << 20);
    hipMalloc(&other, 1 << 16);
    hipMalloc(&sink, NB * 4);
    hipMalloc(&cycles, NB * 16);
    hipMemset(words, 0, 1 << 20);
    hipMemset(other, 0, 1 << 16);
    printf("%6s %4s %7s %5s | header load: median max | load behind the add: median  p90  max (cycles of the 100 MHz.. s_memtime clock as read; us = cycles / 2100 if it ticks at the shader clock)\n",
           "blocks", "W", "stride", "add");
    const int cases[][4] = {{256, 1, 32, 0}, {256, 1, 32, 1}, {512, 1, 32, 0}, {512, 1, 32, 1}, {512, 8, 32, 1}, {512, 16, 32, 1},
                            {512, 16, 64, 1}, {512, 16, 1024, 1}, {512, 64, 32, 1}, {512, 64, 64, 1}, {256, 16, 32, 1}, {256, 16, 64, 1}};
    for (auto& c : cases) {
        std::vector<unsigned long long> h(2 * c[0]), a, b;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(arrive_kernel, dim3(c[0]), dim3(256), 0, 0, words, c[1], c[2], other, c[3], cycles, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cycles, 16 * c[0], hipMemcpyDeviceToHost);
        for (int i = 0; i < c[0]; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        printf("%6d %4d %7d %5d | %6llu %6llu | %6llu %6llu %6llu\n", c[0], c[1], c[2] * 4, c[3], a[a.size() / 2], a.back(),
               b[b.size() / 2], b[b.size() * 9 / 10], b.back());
    }
    return 0;
}
