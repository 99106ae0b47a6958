// ofx_fft_regs.h -- in-register radix-R DFT building blocks for the FUSED engine.
//
// Everything here is fully unrolled at compile time: a thread holds R complex
// values in VGPRs (cpx x[R] with compile-time indices only) and runs a radix-2
// decimation-in-time network on them.  A complex value is a 2-vector in an
// aligned VGPR pair and all arithmetic is packed fp32 (v_pk_add/mul/fma_f32, the
// re<->im swap and the sign pattern of a multiplication by i ride on the
// op_sel / neg modifiers): one instruction per complex operation.  A wave that is
// alone on its SIMD issues one VALU instruction per ~4 cycles, packed or not
// (tools/micro/valu_mix.hip), so halving the instruction count is what counts.
// Twiddles inside a block are compile-time constants (32nd roots of unity); a
// butterfly with a non-trivial twiddle costs 3 packed FMAs (a' = a + w b by two,
// b' = 2a - a' by one), trivial ones 2.  32-point block: 80 butterflies, 194
// instructions for 32 complex points.
#pragma once

#include <hip/hip_runtime.h>

namespace ofxfft {

typedef float cpx __attribute__((ext_vector_type(2)));   // (re, im) in an aligned VGPR pair

__device__ __forceinline__ cpx mk(float re, float im) { return (cpx){re, im}; }
__device__ __forceinline__ cpx swp(cpx z) { return __builtin_shufflevector(z, z, 1, 0); }
__device__ __forceinline__ cpx pfma(cpx a, cpx b, cpx c) { return __builtin_elementwise_fma(a, b, c); }
// The compiler folds whole-vector negation and lane swaps into the packed-instruction
// modifiers but not a negation of one lane; those forms are spelled out here.
//   t + a.y * (-w.y, w.x)
__device__ __forceinline__ cpx pfma_ay_iw(cpx a, cpx w, cpx t) {
    cpx r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
//   t + a.x * (w.x, -w.y)
__device__ __forceinline__ cpx pfma_ax_cw(cpx a, cpx w, cpx t) {
    cpx r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
//   conj(a + b)
__device__ __forceinline__ cpx conj_sum(cpx a, cpx b) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

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
__device__ __forceinline__ void bfly(cpx& a, cpx& b) {
    constexpr int n = ((NUM % 32) + 32) % 32;
    if constexpr (n == 0) {
        const cpx t = b;
        b = a - t;
        a = a + t;
    } else if constexpr (n == 16) {
        const cpx t = b;
        b = a + t;
        a = a - t;
    } else if constexpr (n == 8 || n == 24) {
        // w = +-i  ->  w b = sg * (-b.y, b.x)
        constexpr float sg = (n == 8) ? (float)DIR : (float)-DIR;
        const cpx t = swp(b);
        b = pfma(t, mk(sg, -sg), a);
        a = pfma(t, mk(-sg, sg), a);
    } else {
        constexpr float c = (float)cos32(n);
        constexpr float s = (float)(DIR * sin32(n));
        cpx nn = pfma(b, mk(c, c), a);
        nn = pfma(swp(b), mk(-s, s), nn);
        b = pfma(a, mk(2.0f, 2.0f), -nn);
        a = nn;
    }
}

template <int R, int TOT, int OFF, int LEN, int G, int J, int DIR>
__device__ __forceinline__ void stage_j(cpx (&y)[TOT]) {
    if constexpr (J < LEN / 2) {
        constexpr int bits = ilog2(R);
        // element i of the bit-reversed working array lives in y[OFF + brev(i)]
        bfly<J*(32 / LEN), DIR>(y[OFF + brev(G + J, bits)],
                                 y[OFF + brev(G + J + LEN / 2, bits)]);
        stage_j<R, TOT, OFF, LEN, G, J + 1, DIR>(y);
    }
}
template <int R, int TOT, int OFF, int LEN, int G, int DIR>
__device__ __forceinline__ void stage_g(cpx (&y)[TOT]) {
    if constexpr (G < R) {
        stage_j<R, TOT, OFF, LEN, G, 0, DIR>(y);
        stage_g<R, TOT, OFF, LEN, G + LEN, DIR>(y);
    }
}
template <int R, int TOT, int OFF, int LEN, int DIR>
__device__ __forceinline__ void stages(cpx (&y)[TOT]) {
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
__device__ __forceinline__ void dft(cpx (&x)[TOT]) {
#ifdef ABL_NOFFT      // diagnostic: no butterflies (what the exchanges and the tail cost alone)
    return;
#endif
    constexpr int bits = ilog2(R);
    stages<R, TOT, OFF, 2, DIR>(x);
    cpx t[R];
#pragma unroll
    for (int k = 0; k < R; ++k) t[k] = x[OFF + brev(k, bits)];
#pragma unroll
    for (int k = 0; k < R; ++k) x[OFF + k] = t[k];
}

// w * b for the compile-time constant w = exp(DIR * 2 pi i NUM / 32)
template <int NUM, int DIR>
__device__ __forceinline__ cpx twmul(cpx b) {
    constexpr int n = ((NUM % 32) + 32) % 32;
    if constexpr (n == 0) {
        return b;
    } else if constexpr (n == 16) {
        return -b;
    } else if constexpr (n == 8 || n == 24) {
        constexpr float sg = (n == 8) ? (float)DIR : (float)-DIR;
        return swp(b) * mk(-sg, sg);
    } else {
        constexpr float c = (float)cos32(n);
        constexpr float s = (float)(DIR * sin32(n));
        return pfma(swp(b), mk(-s, s), b * mk(c, c));
    }
}

// complex helpers (two packed instructions each)
__device__ __forceinline__ cpx cmul(cpx a, cpx w) { return pfma_ay_iw(a, w, a.xx * w); }
__device__ __forceinline__ cpx cmulc(cpx a, cpx w) {   // a * conj(w)
    return pfma_ax_cw(a, w, a.yy * swp(w));
}

}  // namespace ofxfft
