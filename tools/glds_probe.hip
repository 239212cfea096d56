// glds_probe.hip -- does global_load_lds_dwordx4 take global addresses that are only 4-byte aligned, and does a partial EXEC mask keep
// the lane-linear destination (base + lane * 16)?  Prints one line per case.  hipcc --offload-arch=gfx950 -O2 tools/glds_probe.hip -o tools/bin/glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k_probe(const int32_t *src, int32_t *out, int shift, int nlanes)
{
    __shared__ __attribute__((aligned(16))) int32_t lds[64 * 4 + 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 + 16; i += 64) lds[i] = -1;
    __syncthreads();
    if (lane < nlanes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + shift + 4 * lane),
                                         (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    __syncthreads();
    for (int i = lane; i < 64 * 4 + 16; i += 64) out[i] = lds[i];
}
int main()
{
    const int N = 4096;
    std::vector<int32_t> h(N);
    for (int i = 0; i < N; i++) h[i] = i * 7 + 3;
    int32_t *d_src, *d_out;
    hipMalloc(&d_src, N * 4); hipMalloc(&d_out, (64 * 4 + 16) * 4);
    hipMemcpy(d_src, h.data(), N * 4, hipMemcpyHostToDevice);
    for (int nl : {64, 33})
        for (int shift = 0; shift < 4; shift++) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_src, d_out, shift, nl);
            std::vector<int32_t> o(64 * 4 + 16);
            if (hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
            int bad = 0, untouched_bad = 0;
            for (int i = 0; i < 4 * nl; i++) bad += o[i] != h[shift + i];
            for (int i = 4 * nl; i < 64 * 4 + 16; i++) untouched_bad += o[i] != -1;
            printf("glds x4: lanes=%d global shift=%d dwords (addr %% 16 = %d): wrong=%d of %d, clobbered beyond=%d\n", nl, shift, (shift * 4) % 16, bad, 4 * nl, untouched_bad);
        }
    return 0;
}
