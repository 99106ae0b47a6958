// micro-benchmark: fp32 VALU issue rate on gfx950 (v_fma_f32 / v_add_f32 / v_pk_fma_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (MODE == 1) x[i] = x[i] + a;
                if (MODE == 2) x[i] = x[i] * a;
            }
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    v2f v = {x[i], x[i + 1]};
                    v2f aa = {a, a}, bb = {b, b};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(aa), "v"(bb));
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
template <int MODE> void run(const char* name, int wgs_per_cu, int lanes_per_instr) {
    int blocks = 256 * wgs_per_cu, iters = 2000;
    float* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 /*waves*/ * iters * 8 * (MODE == 3 ? 8 : 16);
    double per_simd_per_s = instr / (256.0 * 4) / (ms * 1e-3);
    printf("%-14s waves/SIMD=%d  %.3f ms  wave-instr/SIMD/s = %.3e  => cycles/instr @2.4GHz = %.2f ; lane-ops/s = %.3e\n",
           name, wgs_per_cu, ms, per_simd_per_s, 2.4e9 / per_simd_per_s, instr * 64 * (MODE == 3 ? 2 : 1) / (ms * 1e-3));
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, 64);
        run<1>("v_add_f32", w, 64);
        run<2>("v_mul_f32", w, 64);
        run<3>("v_pk_fma_f32", w, 128);
    }
    return 0;
}
