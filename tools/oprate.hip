// VALU issue-rate probe for gfx950: cycles per wave64 instruction of the conversion / f64 ops the descriptor kernel leans on.
// build: hipcc --offload-arch=gfx950 -O2 tools/oprate.hip -o tools/bin/oprate ; run on the GPU box: tools/bin/oprate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 4096
#define OP8(asmtxt, cons)                                                                   \
    for (int i = 0; i < REP; i++) {                                                         \
        asm volatile(asmtxt "\n" asmtxt "\n" asmtxt "\n" asmtxt "\n" asmtxt "\n" asmtxt "\n" asmtxt "\n" asmtxt cons); \
    }
template <int K> __global__ void probe(double *out, unsigned long long *cyc)
{
    double d0 = threadIdx.x * 1.25 + 3.0, d1 = 0; float f0 = threadIdx.x * 0.5f + 1.f, f1 = 0; int i0 = threadIdx.x, i1 = 0;
    float2 p0 = {f0, f0}, p1 = {0, 0};
    unsigned long long t0 = clock64();
    if (K == 0) OP8("v_fma_f64 %0, %1, %1, %1", : "=v"(d1) : "v"(d0));
    if (K == 1) OP8("v_cvt_i32_f64 %0, %1", : "=v"(i1) : "v"(d0));
    if (K == 2) OP8("v_cvt_f32_f64 %0, %1", : "=v"(f1) : "v"(d0));
    if (K == 3) OP8("v_fract_f64 %0, %1", : "=v"(d1) : "v"(d0));
    if (K == 4) OP8("v_cvt_f64_i32 %0, %1", : "=v"(d1) : "v"(i0));
    if (K == 5) OP8("v_mul_lo_u32 %0, %1, %1", : "=v"(i1) : "v"(i0));
    if (K == 6) OP8("v_mad_u32_u24 %0, %1, %1, %1", : "=v"(i1) : "v"(i0));
    if (K == 7) OP8("v_cvt_f32_ubyte0 %0, %1", : "=v"(f1) : "v"(i0));
    if (K == 8) OP8("v_pk_mul_f32 %0, %1, %1", : "=v"(p1) : "v"(p0));
    if (K == 9) OP8("v_rndne_f32 %0, %1", : "=v"(f1) : "v"(f0));
    if (K == 10) OP8("v_add_f64 %0, %1, %1", : "=v"(d1) : "v"(d0));
    if (K == 11) OP8("v_mul_f32 %0, %1, %1", : "=v"(f1) : "v"(f0));
    if (K == 12) OP8("v_cmp_le_f64 vcc, %1, %1", : "=v"(d1) : "v"(d0) : "vcc");
    if (K == 13) OP8("v_rndne_f64 %0, %1", : "=v"(d1) : "v"(d0));
    if (K == 14) OP8("v_lshl_add_u64 %0, %1, 0, %1", : "=v"(d1) : "v"(d0));
    unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d1 + f1 + i1 + p1.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[K] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h[16] = {0};
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, sizeof(h)); hipMemset(cyc, 0, sizeof(h));
    const char *names[] = {"v_fma_f64", "v_cvt_i32_f64", "v_cvt_f32_f64", "v_fract_f64", "v_cvt_f64_i32", "v_mul_lo_u32", "v_mad_u32_u24",
                           "v_cvt_f32_ubyte0", "v_pk_mul_f32", "v_rndne_f32", "v_add_f64", "v_mul_f32", "v_cmp_le_f64", "v_rndne_f64", "v_lshl_add_u64"};
    // one wave per SIMD-ish: 1 block of 64 threads, so the cycle count is the issue cost of a single wave's back-to-back stream
#define RUN(K) hipLaunchKernelGGL(probe<K>, dim3(1), dim3(64), 0, 0, out, cyc);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // clock64 = s_memtime: 100 MHz constant clock on gfx9?  report raw ticks per instruction and relative to v_mul_f32
    for (int k = 0; k < 15; k++) printf("%-18s ticks/inst %.4f  rel %.2f\n", names[k], (double)h[k] / (8.0 * REP), (double)h[k] / (double)h[11]);
    return 0;
}
