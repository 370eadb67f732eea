// Does a buffer_load_dwordx4 / global_load_dwordx4 from an address that is 4-byte but not 16-byte aligned return the right
// four dwords on gfx950?  (The staged weight loaders read fp32 kernels as dwordx4; a parameter view inside a flat buffer
// can start at any multiple of 4 bytes.)  Prints the number of mismatches per misalignment.
// build: hipcc --offload-arch=gfx950 -O3 -o build/unaligned_b128 tools/microbench/unaligned_b128.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* base, int shift, unsigned* out, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(base + shift), 0, n4 * 16, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)i * 16u, 0, 0);
    const u32x4 g = *(const u32x4*)(base + shift + 4 * i);
    out[8 * i + 0] = v.x; out[8 * i + 1] = v.y; out[8 * i + 2] = v.z; out[8 * i + 3] = v.w;
    out[8 * i + 4] = g.x; out[8 * i + 5] = g.y; out[8 * i + 6] = g.z; out[8 * i + 7] = g.w;
}
int main() {
    const int n4 = 4096;
    std::vector<unsigned> h(4 * n4 + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    unsigned *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, n4 * 32);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r(8 * n4);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, d, shift, o, n4);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r.data(), o, n4 * 32, hipMemcpyDeviceToHost);
        int bad_buf = 0, bad_glob = 0;
        for (int i = 0; i < n4; ++i)
            for (int j = 0; j < 4; ++j) {
                bad_buf += r[8 * i + j] != h[shift + 4 * i + j];
                bad_glob += r[8 * i + 4 + j] != h[shift + 4 * i + j];
            }
        printf("base + %d dwords: status %d, buffer_load_dwordx4 mismatches %d, global_load_dwordx4 mismatches %d\n", shift, (int)e, bad_buf, bad_glob);
    }
    return 0;
}
