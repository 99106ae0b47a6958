// micro-benchmark 4: what limits single-wave fp32 VALU issue on gfx950 (instruction mix, operand banks, literals)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define C0 0.70710678f
#define C1 0.92387953f
#define C2 0.38268343f
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16], y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = a * (i + 1) + threadIdx.x; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(y[i])); }
    for (int it = 0; it < iters; ++it) {
#pragma clang loop unroll(full)
        for (int r = 0; r < 64; ++r) {
#pragma clang loop unroll(full)
            for (int i = 0; i < 16; ++i) {
                const int n = r * 16 + i;
                if (MODE == 0) x[i] = x[i] + y[i];                                   // add, same index
                if (MODE == 1) x[i] = x[i] + y[(i + r) & 15];                        // add, rotating operand
                if (MODE == 2) { if (n & 1) x[i] = x[i] + y[i]; else x[i] = __builtin_fmaf(x[i], C0, y[i]); }  // alt add/fmamk
                if (MODE == 3) { if (n & 1) x[i] = x[i] + y[i]; else x[i] = __builtin_fmaf(x[i], a, y[i]); }   // alt add/fma sgpr
                if (MODE == 4) x[i] = __builtin_fmaf(x[i], (n % 3 == 0 ? C0 : n % 3 == 1 ? C1 : C2), y[i]);    // fmamk 3 literals
                if (MODE == 5) x[i] = __builtin_fmaf(x[i], C0, y[(i + r) & 15]);     // fmamk rotating operand
                if (MODE == 6) { if (n & 1) x[i] = x[i] - y[i]; else x[i] = x[i] + y[i]; }  // alt add/sub
                if (MODE == 8) { if (n & 1) asm("v_fma_f32 %0, %1, 1.0, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]));
                                 else asm("v_fma_f32 %0, %1, 1.0, -%2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i])); }      // fma add / fma sub
                if (MODE == 9) { if (n & 1) asm("v_fma_f32 %0, %1, %3, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]), "s"(a));
                                 else asm("v_fma_f32 %0, %1, %3, -%2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]), "s"(b)); }  // fma two sgprs
                if (MODE == 10) { if (n & 1) asm("v_fma_f32 %0, %1, 1.0, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]));
                                  else asm("v_fma_f32 %0, %1, %3, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]), "s"(a)); }  // fma inline const / sgpr
                if (MODE == 11) { if (n & 1) asm("v_fma_f32 %0, %1, 1.0, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]));
                                  else asm("v_fma_f32 %0, %1, %3, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]), "v"(y[(i+5)&15])); }  // fma const / 3 vgpr
                if (MODE == 12) { if (n & 1) asm("v_fma_f32 %0, %1, 1.0, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]));
                                  else asm("v_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i])); }  // fma / mul
                if (MODE == 13) { if (n & 1) asm("v_pk_add_f32 %0, %1, %2" : "=v"(*(v2f*)&x[i & 14]) : "v"(*(v2f*)&x[i & 14]), "v"(*(v2f*)&y[i & 14]));
                                  else asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(*(v2f*)&x[i & 14]) : "v"(*(v2f*)&x[i & 14]), "v"(*(v2f*)&y[i & 14])); }  // pk add / pk sub
                if (MODE == 7) { if ((n >> 2) & 1) x[i] = x[i] + y[i]; else x[i] = __builtin_fmaf(x[i], C0, y[i]); } // runs of 4
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int wgs_per_cu) {
    int blocks = 256 * wgs_per_cu, iters = 512;
    float* d; (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 2, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 * iters * 1024;
    printf("%-28s waves/SIMD=%d  ns/instr/SIMD = %.3f\n", name, wgs_per_cu, 1e9 / (instr / 1024.0 / (ms * 1e-3)));
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2}) {
        run<0>("add same idx", w); run<1>("add rotating", w); run<2>("alt add/fmamk", w); run<3>("alt add/fma-sgpr", w);
        run<4>("fmamk 3 literals", w); run<5>("fmamk rotating", w); run<6>("alt add/sub", w); run<7>("runs of 4 add/fmamk", w);
        run<8>("fma +y / fma -y", w); run<9>("fma s0 / fma s1", w); run<10>("fma 1.0 / fma sgpr", w); run<11>("fma 1.0 / fma vvv", w);
        run<12>("fma / mul", w); run<13>("pk_add / pk_add neg", w);
    }
    return 0;
}
