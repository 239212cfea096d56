// VALU issue-THROUGHPUT calibration for gfx950 (MI355X): settles whether a wave64 VALU instruction occupies its SIMD for 4 cycles
// (16 lanes / clock: 256 CUs x 4 SIMDs x 16 x 2.4 GHz = 39.3 T lane-ops/s, bench.py's VALU_PEAK_TLANEOPS) or for 2 (78.6 T).
// tools/oprate.hip times ONE wave (issue latency); this fills every SIMD of the chip with 1 / 2 / 4 / 8 waves, each running 8
// independent dependency chains of one instruction, and reports lane-operations per second from HIP-event times.
//   build: hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o tools/bin/valu_peak ; run on the GPU box: tools/bin/valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <dirent.h>
#include <string.h>
#include <string>
#include <unistd.h>
#define REP 2048            // loop trips per outer round; 8 instructions per trip.  Rounds are sized at run time so that every point lasts >= 20 ms
#define CHAINS8(OP, C, V)                                                               \
    for (int i = 0; i < REP * outer; i++) {                                                     \
        asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
                     OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8"   \
                     : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]) : "v"(C)); \
    }
#define CHAINS8_3(OP, C, V)                                                             \
    for (int i = 0; i < REP * outer; i++) {                                                     \
        asm volatile(OP " %0, %0, %8, %8\n" OP " %1, %1, %8, %8\n" OP " %2, %2, %8, %8\n" OP " %3, %3, %8, %8\n" \
                     OP " %4, %4, %8, %8\n" OP " %5, %5, %8, %8\n" OP " %6, %6, %8, %8\n" OP " %7, %7, %8, %8"   \
                     : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]) : "v"(C)); \
    }
#define CHAINS8_CVT(OP, V, U)                                                           \
    for (int i = 0; i < REP * outer; i++) {                                                     \
        asm volatile(OP " %0, %8\n" OP " %1, %9\n" OP " %2, %10\n" OP " %3, %11\n"      \
                     OP " %4, %12\n" OP " %5, %13\n" OP " %6, %14\n" OP " %7, %15"      \
                     : "=v"(V[0]), "=v"(V[1]), "=v"(V[2]), "=v"(V[3]), "=v"(V[4]), "=v"(V[5]), "=v"(V[6]), "=v"(V[7])            \
                     : "v"(U[0]), "v"(U[1]), "v"(U[2]), "v"(U[3]), "v"(U[4]), "v"(U[5]), "v"(U[6]), "v"(U[7]));                 \
    }

template <int K> __global__ void __launch_bounds__(256) probe(float *out, int outer)
{
    const float c = 1.0000001f;
    float f[8]; double d[8]; float2 p[8]; unsigned u[8]; int s[8];
    for (int k = 0; k < 8; k++) { f[k] = threadIdx.x * 0.5f + k; d[k] = threadIdx.x * 1.25 + k; p[k] = make_float2(f[k], f[k] + 1.f); u[k] = threadIdx.x * 2654435761u + k; s[k] = (int)u[k]; }
    const double cd = 1.0000000001; const float2 cp = make_float2(c, c); const int ci = 3;
    if (K == 0) CHAINS8("v_mul_f32", c, f)
    if (K == 1) CHAINS8_3("v_fma_f32", c, f)
    if (K == 2) CHAINS8("v_pk_mul_f32", cp, p)
    if (K == 3) CHAINS8_3("v_fma_f64", cd, d)
    if (K == 4) CHAINS8("v_add_f64", cd, d)
    if (K == 5) CHAINS8_CVT("v_cvt_f32_ubyte0", f, u)
    if (K == 6) CHAINS8_CVT("v_cvt_f64_i32", d, s)
    if (K == 7) CHAINS8_CVT("v_cvt_i32_f64", s, d)
    if (K == 8) CHAINS8("v_add_u32", ci, s)
    if (K == 9) CHAINS8_3("v_mad_u32_u24", ci, s)
    if (K == 10) CHAINS8_CVT("v_fract_f64", d, d)
    if (K == 11) CHAINS8_CVT("v_cvt_f32_f64", f, d)
    float acc = 0;
    for (int k = 0; k < 8; k++) acc += f[k] + (float)d[k] + p[k].x + p[k].y + (float)s[k];
    if (acc == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;      // never true: keeps the chains alive
}

typedef void (*kern_t)(float *, int);

// current shader clock (MHz) from sysfs: the starred line of pp_dpm_sclk of the first card that has one; read WHILE the probe runs
static int sclk_mhz()
{
    DIR *d = opendir("/sys/class/drm");
    if (!d) return 0;
    int best = 0;
    while (struct dirent *e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) || strchr(e->d_name, '-')) continue;
        std::string p = std::string("/sys/class/drm/") + e->d_name + "/device/pp_dpm_sclk";
        FILE *f = fopen(p.c_str(), "r");
        if (!f) continue;
        char line[128];
        while (fgets(line, sizeof(line), f))
            if (strchr(line, '*')) { int lvl = 0, mhz = 0; if (sscanf(line, "%d: %dMhz", &lvl, &mhz) == 2 && mhz > best) best = mhz; }
        fclose(f);
        if (best) break;
    }
    closedir(d);
    return best;
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("# device %s, %d CUs, clockRate attribute %.0f MHz\n", prop.gcnArchName, cus, khz / 1e3);
    printf("# lane-ops/s = waves x %d trips x 8 instructions x 64 lanes / HIP-event time; waves/SIMD = resident waves per SIMD (256-thread workgroups = 1 wave per SIMD each)\n", REP);
    float *out; hipMalloc(&out, 64 << 20);
    const char *names[] = {"v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_fma_f64", "v_add_f64", "v_cvt_f32_ubyte0", "v_cvt_f64_i32", "v_cvt_i32_f64",
                           "v_add_u32", "v_mad_u32_u24", "v_fract_f64", "v_cvt_f32_f64"};
    kern_t kerns[] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>, probe<11>};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // one second of load first: the part needs about that long to settle on its sustained clock
    for (int r = 0; r < 40; r++) hipLaunchKernelGGL(kerns[1], dim3(cus * 8), dim3(256), 0, 0, out, 110);
    hipDeviceSynchronize();
    printf("# every point: >= 20 ms of kernel time (rounds sized from a calibration launch), after ~1 s of load; sclk = the starred level of pp_dpm_sclk read while the probe runs\n");
    for (int k = 0; k < 12; k++) {
        printf("%-18s", names[k]);
        for (int wps = 1; wps <= 8; wps *= 2) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(kerns[k], dim3(cus * wps), dim3(256), 0, 0, out, 4);      // calibration: 4 rounds
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms4; hipEventElapsedTime(&ms4, a, b);
            int outer = (int)(20.0 / (ms4 / 4.0)) + 1;
            if (outer < 8) outer = 8;
            double best = 1e30; int clk = 0;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(kerns[k], dim3(cus * wps), dim3(256), 0, 0, out, outer);
                hipEventRecord(b, 0);
                usleep(8000);
                const int c = sclk_mhz();
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) { best = ms; clk = c; }
            }
            const double waves = (double)cus * wps * 4;
            const double laneops = waves * REP * (double)outer * 8.0 * 64.0;
            const double rate = laneops / (best * 1e-3) / 1e12;
            const double ghz = clk ? clk / 1e3 : 2.4;
            printf("  w/SIMD=%d: %7.2f T lane-ops/s (%.1f ms, sclk %d MHz, %.2f cyc/inst/SIMD @sclk)", wps, rate, best, clk,
                   (best * 1e-3 * ghz * 1e9) / (wps * REP * (double)outer * 8.0));
        }
        printf("\n");
    }
    return 0;
}
