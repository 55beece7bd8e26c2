// valu_issue.hip -- how many VALU wave-instructions per cycle does one gfx950 SIMD issue, as a function of the
// waves resident on it?  Settles how to read SQ_ACTIVE_INST_VALU (quad-cycle units) for the collide kernel:
// is "busy" = instructions x 4 cycles (one wave alone) or x 2 (SIMD-32 rate with several waves)?
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int KIND>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *cyc, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v0 = {a0, a1}, v1 = {a2, a3}, v2 = {a4, a5}, v3 = {a6, a7}, v4 = {a1, a0}, v5 = {a3, a2}, v6 = {a5, a4}, v7 = {a7, a6};
    const f2 m2 = {m, m}, c2 = {c, c};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {          // independent FMAs (ffp-contract on by default for this probe)
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            } else if (KIND == 1) {   // integer add / xor mix
                i0 = (i0 + i1) ^ 0x55; i1 = (i1 + i2) ^ 0x33; i2 = (i2 + i3) ^ 0x0f; i3 = (i3 + i4) ^ 0x11;
                i4 = (i4 + i5) ^ 0x77; i5 = (i5 + i6) ^ 0x13; i6 = (i6 + i7) ^ 0x17; i7 = (i7 + i0) ^ 0x19;
            } else if (KIND == 2) {   // compare + select
                a0 = a0 > a1 ? a2 : a0; a1 = a1 > a2 ? a3 : a1; a2 = a2 > a3 ? a4 : a2; a3 = a3 > a4 ? a5 : a3;
                a4 = a4 > a5 ? a6 : a4; a5 = a5 > a6 ? a7 : a5; a6 = a6 > a7 ? a0 : a6; a7 = a7 > a0 ? a1 : a7;
            } else if (KIND == 4) {   // independent packed FMAs (v_pk_fma_f32: two f32 per lane)
                v0 = __builtin_elementwise_fma(v0, m2, c2); v1 = __builtin_elementwise_fma(v1, m2, c2); v2 = __builtin_elementwise_fma(v2, m2, c2); v3 = __builtin_elementwise_fma(v3, m2, c2);
                v4 = __builtin_elementwise_fma(v4, m2, c2); v5 = __builtin_elementwise_fma(v5, m2, c2); v6 = __builtin_elementwise_fma(v6, m2, c2); v7 = __builtin_elementwise_fma(v7, m2, c2);
            } else if (KIND == 5) {   // independent packed multiplies then adds (v_pk_mul_f32, v_pk_add_f32; 16 instructions)
                v0 = v0 * m2; v1 = v1 * m2; v2 = v2 * m2; v3 = v3 * m2; v4 = v4 * m2; v5 = v5 * m2; v6 = v6 * m2; v7 = v7 * m2;
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
                v0 = v0 + c2; v1 = v1 + c2; v2 = v2 + c2; v3 = v3 + c2; v4 = v4 + c2; v5 = v5 + c2; v6 = v6 + c2; v7 = v7 + c2;
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            } else if (KIND == 6) {   // one dependent chain of packed FMAs (latency)
                v0 = __builtin_elementwise_fma(v0, m2, c2); v0 = __builtin_elementwise_fma(v0, m2, c2); v0 = __builtin_elementwise_fma(v0, m2, c2); v0 = __builtin_elementwise_fma(v0, m2, c2);
                v0 = __builtin_elementwise_fma(v0, m2, c2); v0 = __builtin_elementwise_fma(v0, m2, c2); v0 = __builtin_elementwise_fma(v0, m2, c2); v0 = __builtin_elementwise_fma(v0, m2, c2);
            } else {                  // one dependent chain (latency)
                a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
                a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
            }
        }
    }
    const long long t1 = clock64();
    a0 += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y + v4.x + v4.y + v5.x + v5.y + v6.x + v6.y + v7.x + v7.y;
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
}

template <int KIND>
static void run(const char *name, int waves_per_simd, int per_iter_instr)
{
    const int cus = 256, iters = 20000;
    const int blocks = cus * 4 * waves_per_simd;        // 64-thread blocks: one wave each
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, 64>>>(out, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<blocks, 64>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = (unsigned long long *)malloc(blocks * 8);
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    const double instr = (double)iters * 8 * per_iter_instr;       // per wave
    // clock64 ticks at 100 MHz on gfx9 (s_memrealtime)?  report both views
    printf("%-10s waves/SIMD %d: %8.3f ms wall, clock64 delta/wave %.0f -> %.2f ticks per wave-instr; "
           "SIMD issue interval (wall x 2.4 GHz / instr per SIMD) %.2f cycles\n", name, waves_per_simd, ms, mean,
           mean / instr, ms * 1e-3 * 2.4e9 / (instr * waves_per_simd));
    hipFree(out); hipFree(cyc); free(h);
}

int main()
{
    for (int w : {1, 2, 4, 8}) run<0>("fma", w, 8);
    for (int w : {1, 2, 4, 8}) run<1>("iadd+xor", w, 16);
    for (int w : {1, 2, 4, 8}) run<2>("cmp+sel", w, 16);
    for (int w : {1, 2, 4, 8}) run<3>("dep-fma", w, 8);
    for (int w : {1, 2, 4, 8}) run<4>("pk_fma", w, 8);
    for (int w : {1, 2, 4, 8}) run<5>("pk_mul+add", w, 16);
    for (int w : {1, 2, 4, 8}) run<6>("dep-pk_fma", w, 8);
    return 0;
}
