// micro-benchmark 2: fp32 VALU issue rate vs. number of VGPR source operands (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16], y[16], z[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = a * (i + 1) + threadIdx.x; z[i] = b * (i + 2) - threadIdx.x; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(y[i]), "+v"(z[i])); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);              // 1 VGPR + 2 SGPR
                if (MODE == 1) x[i] = x[i] + y[i];                             // 2 VGPR
                if (MODE == 2) x[i] = __builtin_fmaf(x[i], y[i], z[i]);        // 3 VGPR
                if (MODE == 3) x[i] = __builtin_fmaf(x[i], 0.70710678f, y[i]); // fmamk literal + 2 VGPR
                if (MODE == 4) x[i] = __builtin_fmaf(x[i], a, y[i]);           // 2 VGPR + SGPR
                if (MODE == 7) x[i] = x[i] + y[(i + 4) & 15];                   // 2 VGPR other bank pattern
            }
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    v2f v = {x[i], x[i + 1]};
                    v2f yy = {y[i], y[i + 1]};
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(yy));
                    x[i] = v.x; x[i + 1] = v.y;
                }
            }
            if (MODE == 6) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    v2f v = {x[i], x[i + 1]};
                    v2f yy = {y[i], y[i + 1]}, zz = {z[i], z[i + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(yy), "v"(zz));
                    x[i] = v.x; x[i + 1] = v.y;
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int wgs_per_cu) {
    int blocks = 256 * wgs_per_cu, iters = 4000;
    float* d; (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const bool pk = (MODE == 5 || MODE == 6);
    double instr = (double)blocks * 4 * iters * 8 * (pk ? 8 : 16);
    double per_simd_per_s = instr / (256.0 * 4) / (ms * 1e-3);
    printf("%-22s waves/SIMD=%d  %.3f ms  ns/instr/SIMD = %.3f  lane-ops/s = %.3e\n", name, wgs_per_cu, ms,
           1e9 / per_simd_per_s, instr * 64 * (pk ? 2 : 1) / (ms * 1e-3));
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("fma v,s,s", w);
        run<1>("add v,v", w);
        run<7>("add v,v(other bank)", w);
        run<2>("fma v,v,v", w);
        run<3>("fmamk v,lit,v", w);
        run<4>("fma v,s,v", w);
        run<5>("pk_add v,v", w);
        run<6>("pk_fma v,v,v", w);
    }
    return 0;
}
