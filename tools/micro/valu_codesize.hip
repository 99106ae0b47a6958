// micro-benchmark 3: fp32 VALU issue rate vs. loop-body code size (instruction fetch), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int REP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16], y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = a * (i + 1) + threadIdx.x; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(y[i])); }
    for (int it = 0; it < iters; ++it) {
#pragma clang loop unroll(full)
        for (int r = 0; r < REP; ++r) {
#pragma clang loop unroll(full)
            for (int i = 0; i < 16; ++i) {
                // alternate 4-byte (v_add e32) and 8-byte (v_fmamk literal) encodings like the DFT code
                if ((i + r) & 1) x[i] = x[i] + y[(i + r) & 15];
                else x[i] = __builtin_fmaf(x[i], (((r * 16 + i) % 3) == 0 ? 0.70710678f : ((r * 16 + i) % 3) == 1 ? 0.92387953f : 0.38268343f), y[(i + 3 * r) & 15]);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int REP> void run(int wgs_per_cu) {
    int blocks = 256 * wgs_per_cu;
    int iters = 32768 / REP;
    float* d; (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<REP><<<blocks, 256>>>(d, 2, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<REP><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 * iters * REP * 16;
    double per_simd_per_s = instr / (256.0 * 4) / (ms * 1e-3);
    printf("body=%5d instr (~%3d KB) waves/SIMD=%d  %.3f ms  ns/instr/SIMD = %.3f\n", REP * 16, REP * 16 * 6 / 1024,
           wgs_per_cu, ms, 1e9 / per_simd_per_s);
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2}) {
        run<8>(w); run<64>(w); run<256>(w); run<512>(w); run<640>(w); run<768>(w); run<1024>(w);
    }
    return 0;
}
