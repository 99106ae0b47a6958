// ofx_fft_regs.h -- in-register radix-R DFT building blocks for the FUSED engine.
//
// Everything here is fully unrolled at compile time: a thread holds R complex
// values in VGPRs (float2 x[R] with compile-time indices only) and runs a
// radix-2 decimation-in-time network on them.  Twiddles inside a block are
// compile-time constants (32nd roots of unity); a butterfly with a non-trivial
// twiddle costs 6 FMAs (a' = a + w b by 4 FMAs, b' = 2a - a' by 2), trivial
// ones 4 adds.  32-point block: 46 trivial + 34 general butterflies = 388 VALU
// instructions for 32 complex points.
#pragma once

#include <hip/hip_runtime.h>

namespace ofxfft {

// cos(2 pi j / 32), j = 0..8
constexpr double kCos32[9] = {
    1.0,
    0.98078528040323044913,
    0.92387953251128675613,
    0.83146961230254523708,
    0.70710678118654752440,
    0.55557023301960222474,
    0.38268343236508977173,
    0.19509032201612826785,
    0.0};

__host__ __device__ constexpr double cos32(int j) {
    j = ((j % 32) + 32) % 32;
    if (j <= 8) return kCos32[j];
    if (j <= 16) return -kCos32[16 - j];
    if (j <= 24) return -kCos32[j - 16];
    return kCos32[32 - j];
}
__host__ __device__ constexpr double sin32(int j) { return cos32(j - 8); }

__host__ __device__ constexpr int brev(int v, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b)
        if (v & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}
__host__ __device__ constexpr int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// One radix-2 DIT butterfly with the constant twiddle w = exp(DIR * 2 pi i NUM/32):
//   a' = a + w b ,  b' = a - w b.
template <int NUM, int DIR>
__device__ __forceinline__ void bfly(float2& a, float2& b) {
    constexpr int n = ((NUM % 32) + 32) % 32;
    if constexpr (n == 0) {
        const float2 t = b;
        b = make_float2(a.x - t.x, a.y - t.y);
        a = make_float2(a.x + t.x, a.y + t.y);
    } else if constexpr (n == 8) {
        // w = DIR * i  ->  w b = DIR * (-b.y, b.x)
        const float2 t = (DIR > 0) ? make_float2(-b.y, b.x) : make_float2(b.y, -b.x);
        b = make_float2(a.x - t.x, a.y - t.y);
        a = make_float2(a.x + t.x, a.y + t.y);
    } else if constexpr (n == 16) {
        const float2 t = b;
        b = make_float2(a.x + t.x, a.y + t.y);
        a = make_float2(a.x - t.x, a.y - t.y);
    } else if constexpr (n == 24) {
        const float2 t = (DIR > 0) ? make_float2(b.y, -b.x) : make_float2(-b.y, b.x);
        b = make_float2(a.x - t.x, a.y - t.y);
        a = make_float2(a.x + t.x, a.y + t.y);
    } else {
        constexpr float c = (float)cos32(n);
        constexpr float s = (float)(DIR * sin32(n));
        const float nx = fmaf(c, b.x, fmaf(-s, b.y, a.x));
        const float ny = fmaf(c, b.y, fmaf(s, b.x, a.y));
        b = make_float2(fmaf(2.0f, a.x, -nx), fmaf(2.0f, a.y, -ny));
        a = make_float2(nx, ny);
    }
}

template <int R, int TOT, int OFF, int LEN, int G, int J, int DIR>
__device__ __forceinline__ void stage_j(float2 (&y)[TOT]) {
    if constexpr (J < LEN / 2) {
        constexpr int bits = ilog2(R);
        // element i of the bit-reversed working array lives in y[OFF + brev(i)]
        bfly<J*(32 / LEN), DIR>(y[OFF + brev(G + J, bits)],
                                 y[OFF + brev(G + J + LEN / 2, bits)]);
        stage_j<R, TOT, OFF, LEN, G, J + 1, DIR>(y);
    }
}
template <int R, int TOT, int OFF, int LEN, int G, int DIR>
__device__ __forceinline__ void stage_g(float2 (&y)[TOT]) {
    if constexpr (G < R) {
        stage_j<R, TOT, OFF, LEN, G, 0, DIR>(y);
        stage_g<R, TOT, OFF, LEN, G + LEN, DIR>(y);
    }
}
template <int R, int TOT, int OFF, int LEN, int DIR>
__device__ __forceinline__ void stages(float2 (&y)[TOT]) {
    if constexpr (LEN <= R) {
        stage_g<R, TOT, OFF, LEN, 0, DIR>(y);
        stages<R, TOT, OFF, LEN * 2, DIR>(y);
    }
}

// In-place R-point DFT, natural order in and out:
//   x[k] <- sum_j x[j] exp(DIR * 2 pi i j k / R),   R in {2,4,8,16,32}.
// Working array w[i] (DIT, input bit-reversed): w[i] = x[brev(i)]; we never move
// data -- stage_j addresses w[i] as x[brev(i)] -- and after the last stage the
// natural-order output k sits in w[k] = x[brev(k)], so one compile-time
// register renaming at the end restores natural order.
// Operates on x[OFF .. OFF+R) of an array of TOT registers.
template <int R, int DIR, int TOT = R, int OFF = 0>
__device__ __forceinline__ void dft(float2 (&x)[TOT]) {
    constexpr int bits = ilog2(R);
    stages<R, TOT, OFF, 2, DIR>(x);
    float2 t[R];
#pragma unroll
    for (int k = 0; k < R; ++k) t[k] = x[OFF + brev(k, bits)];
#pragma unroll
    for (int k = 0; k < R; ++k) x[OFF + k] = t[k];
}

// complex helpers
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 w) {   // a * conj(w)
    return make_float2(fmaf(a.x, w.x, a.y * w.y), fmaf(a.y, w.x, -a.x * w.y));
}

}  // namespace ofxfft
