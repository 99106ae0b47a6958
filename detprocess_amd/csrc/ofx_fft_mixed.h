// ofx_fft_mixed.h -- in-register DFT blocks of 20 and 25 points for the register-resident
// kernel of the reference example's 25000-sample traces (ofx_fused25.hip;
// /root/reference/examples/processing/process_example.yaml:93).
//
// Same conventions as ofx_fft_regs.h: a thread holds the points in VGPR pairs (compile-time
// indices only), every complex operation is one packed fp32 instruction, twiddles inside a
// block are compile-time constants (50th roots of unity here).
//   radix-5 butterfly   18 instructions    radix-4 butterfly   8 instructions
//   dft25 = 5 x 5 Cooley-Tukey: 10 radix-5 butterflies + 16 constant twiddles = 212
//   dft20 = 4 x 5 prime-factor (Good-Thomas) map: no twiddles, 5 radix-4 + 4 radix-5 = 112
#pragma once

#include "ofx_fft_regs.h"

namespace ofxfft {

constexpr double kCos50[50] = {
    1.00000000000000000000, 0.99211470131447787590, 0.96858316112863107605, 0.92977648588825145826,
    0.87630668004386358394, 0.80901699437494745126, 0.72896862742141155245, 0.63742398974868974548,
    0.53582679497899654564, 0.42577929156507265951, 0.30901699437494745126, 0.18738131458572473975,
    0.06279051952931352654, -0.06279051952931340164, -0.18738131458572460097, -0.30901699437494711820,
    -0.42577929156507271502, -0.53582679497899687870, -0.63742398974868974548, -0.72896862742141133040,
    -0.80901699437494734024, -0.87630668004386358394, -0.92977648588825134723, -0.96858316112863096503,
    -0.99211470131447776488, -1.00000000000000000000, -0.99211470131447787590, -0.96858316112863118708,
    -0.92977648588825145826, -0.87630668004386347292, -0.80901699437494778433, -0.72896862742141177449,
    -0.63742398974868952344, -0.53582679497899632359, -0.42577929156507215991, -0.30901699437494756229,
    -0.18738131458572462873, -0.06279051952931320735, 0.06279051952931283265, 0.18738131458572426791,
    0.30901699437494722922, 0.42577929156507182684, 0.53582679497899676768, 0.63742398974868930139,
    0.72896862742141121938, 0.80901699437494734024, 0.87630668004386313985, 0.92977648588825145826,
    0.96858316112863096503, 0.99211470131447776488};
constexpr double kSin50[50] = {
    0.00000000000000000000, 0.12533323356430425832, 0.24868988716485479484, 0.36812455268467791925,
    0.48175367410171532345, 0.58778525229247313710, 0.68454710592868861507, 0.77051324277578925326,
    0.84432792550201507531, 0.90482705246601957683, 0.95105651629515353118, 0.98228725072868861012,
    0.99802672842827155897, 0.99802672842827155897, 0.98228725072868872115, 0.95105651629515364220,
    0.90482705246601946580, 0.84432792550201496429, 0.77051324277578925326, 0.68454710592868883712,
    0.58778525229247324813, 0.48175367410171521243, 0.36812455268467814129, 0.24868988716485523893,
    0.12533323356430453588, 0.0, -0.12533323356430428608, -0.24868988716485457280,
    -0.36812455268467791925, -0.48175367410171537896, -0.58778525229247269301, -0.68454710592868839303,
    -0.77051324277578936428, -0.84432792550201529735, -0.90482705246601979887, -0.95105651629515353118,
    -0.98228725072868872115, -0.99802672842827155897, -0.99802672842827155897, -0.98228725072868872115,
    -0.95105651629515364220, -0.90482705246601990989, -0.84432792550201496429, -0.77051324277578958633,
    -0.68454710592868894814, -0.58778525229247335915, -0.48175367410171610061, -0.36812455268467786373,
    -0.24868988716485534995, -0.12533323356430464690};
__host__ __device__ constexpr double cos50(int j) { return kCos50[((j % 50) + 50) % 50]; }
__host__ __device__ constexpr double sin50(int j) { return kSin50[((j % 50) + 50) % 50]; }

// b * exp(DIR * 2 pi i NUM / 50) for a compile-time NUM
template <int NUM, int DIR>
__device__ __forceinline__ cpx twmul50(cpx b) {
    constexpr int n = ((NUM % 50) + 50) % 50;
    if constexpr (n == 0) {
        return b;
    } else if constexpr (n == 25) {
        return -b;
    } else {
        constexpr float c = (float)cos50(n);
        constexpr float s = (float)(DIR * sin50(n));
        return pfma(swp(b), mk(-s, s), b * mk(c, c));
    }
}

// 5-point DFT, x_k <- sum_j x_j exp(DIR 2 pi i j k / 5), in place on five references
template <int DIR>
__device__ __forceinline__ void r5(cpx& x0, cpx& x1, cpx& x2, cpx& x3, cpx& x4) {
    constexpr float c1 = (float)cos50(10), c2 = (float)cos50(20);
    constexpr float s1 = (float)sin50(10), s2 = (float)sin50(20);
    constexpr float dr = (float)DIR;
    const cpx t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    const cpx a1 = pfma(t2, mk(c2, c2), pfma(t1, mk(c1, c1), x0));
    const cpx a2 = pfma(t2, mk(c1, c1), pfma(t1, mk(c2, c2), x0));
    const cpx b1 = pfma(t4, mk(s2, s2), t3 * mk(s1, s1));
    const cpx b2 = pfma(t4, mk(-s1, -s1), t3 * mk(s2, s2));
    x0 = (x0 + t1) + t2;
    // y1 = a1 + DIR i b1, y4 = a1 - DIR i b1, y2 = a2 + DIR i b2, y3 = a2 - DIR i b2
    const cpx sb1 = swp(b1), sb2 = swp(b2);
    x1 = pfma(sb1, mk(-dr, dr), a1);
    x4 = pfma(sb1, mk(dr, -dr), a1);
    x2 = pfma(sb2, mk(-dr, dr), a2);
    x3 = pfma(sb2, mk(dr, -dr), a2);
}

// 4-point DFT in place
template <int DIR>
__device__ __forceinline__ void r4(cpx& x0, cpx& x1, cpx& x2, cpx& x3) {
    constexpr float dr = (float)DIR;
    const cpx t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3;
    const cpx t3 = swp(x1 - x3);
    x0 = t0 + t2;
    x2 = t0 - t2;
    x1 = pfma(t3, mk(-dr, dr), t1);
    x3 = pfma(t3, mk(dr, -dr), t1);
}

// ---- 25 points: input index n = 5 a + b, output index k = c + 5 e
//   step 1 (per b): radix-5 over a on positions {5 a + b}      -> u_b[c] at position 5 c + b
//   step 2: position 5 c + b  *=  w_25^{b c}
//   step 3 (per c): radix-5 over b on positions {5 c + b}      -> y[c + 5 e] at position 5 c + e
//   renaming: natural-order output k = c + 5 e is read from position 5 c + e
template <int DIR, int TOT, int OFF, int B>
__device__ __forceinline__ void d25_s1(cpx (&x)[TOT]) {
    if constexpr (B < 5) {
        r5<DIR>(x[OFF + B], x[OFF + 5 + B], x[OFF + 10 + B], x[OFF + 15 + B], x[OFF + 20 + B]);
        d25_s1<DIR, TOT, OFF, B + 1>(x);
    }
}
template <int DIR, int TOT, int OFF, int I>
__device__ __forceinline__ void d25_tw(cpx (&x)[TOT]) {
    if constexpr (I < 25) {
        constexpr int c = I / 5, b = I % 5;
        if constexpr (b * c != 0) x[OFF + I] = twmul50<2 * b * c, DIR>(x[OFF + I]);
        d25_tw<DIR, TOT, OFF, I + 1>(x);
    }
}
template <int DIR, int TOT, int OFF, int C>
__device__ __forceinline__ void d25_s3(cpx (&x)[TOT]) {
    if constexpr (C < 5) {
        r5<DIR>(x[OFF + 5 * C], x[OFF + 5 * C + 1], x[OFF + 5 * C + 2], x[OFF + 5 * C + 3],
                x[OFF + 5 * C + 4]);
        d25_s3<DIR, TOT, OFF, C + 1>(x);
    }
}
template <int DIR, int TOT, int OFF>
__device__ __forceinline__ void dft25(cpx (&x)[TOT]) {
#ifdef ABL_NOFFT
    return;
#endif
    d25_s1<DIR, TOT, OFF, 0>(x);
    d25_tw<DIR, TOT, OFF, 0>(x);
    d25_s3<DIR, TOT, OFF, 0>(x);
    cpx t[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) t[k] = x[OFF + 5 * (k % 5) + k / 5];
#pragma unroll
    for (int k = 0; k < 25; ++k) x[OFF + k] = t[k];
}

// ---- 20 points, prime-factor map (4 and 5 are coprime: no twiddles)
//   input  n = (5 n1 + 4 n2) mod 20      output k = (5 k1 + 16 k2) mod 20
//   step 1 (per n2): radix-4 over n1 on positions (5 n1 + 4 n2) mod 20, result k1 in place
//   step 2 (per k1): radix-5 over n2 on positions (5 k1 + 4 n2) mod 20, result k2 in place
//   renaming: output k = (5 k1 + 16 k2) mod 20 is read from position (5 k1 + 4 k2) mod 20
__host__ __device__ constexpr int p20(int a, int b) { return (5 * a + 4 * b) % 20; }
template <int DIR, int TOT, int OFF, int N2>
__device__ __forceinline__ void d20_s1(cpx (&x)[TOT]) {
    if constexpr (N2 < 5) {
        r4<DIR>(x[OFF + p20(0, N2)], x[OFF + p20(1, N2)], x[OFF + p20(2, N2)], x[OFF + p20(3, N2)]);
        d20_s1<DIR, TOT, OFF, N2 + 1>(x);
    }
}
template <int DIR, int TOT, int OFF, int K1>
__device__ __forceinline__ void d20_s2(cpx (&x)[TOT]) {
    if constexpr (K1 < 4) {
        r5<DIR>(x[OFF + p20(K1, 0)], x[OFF + p20(K1, 1)], x[OFF + p20(K1, 2)], x[OFF + p20(K1, 3)],
                x[OFF + p20(K1, 4)]);
        d20_s2<DIR, TOT, OFF, K1 + 1>(x);
    }
}
__host__ __device__ constexpr int d20_src(int k) {
    // k = (5 k1 + 16 k2) mod 20:  k1 = k mod 4 (16 k2 = 0 mod 4, 5 = 1 mod 4),  k2 = k mod 5 (16 = 1)
    return p20(k % 4, k % 5);
}
template <int DIR, int TOT, int OFF>
__device__ __forceinline__ void dft20(cpx (&x)[TOT]) {
#ifdef ABL_NOFFT
    return;
#endif
    d20_s1<DIR, TOT, OFF, 0>(x);
    d20_s2<DIR, TOT, OFF, 0>(x);
    cpx t[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) t[k] = x[OFF + d20_src(k)];
#pragma unroll
    for (int k = 0; k < 20; ++k) x[OFF + k] = t[k];
}

// ---- 10 points, prime-factor map 2 x 5 (ofx_fused25.hip built for 12500-sample traces)
//   input  n = (5 n1 + 2 n2) mod 10      output k = (5 k1 + 6 k2) mod 10
__host__ __device__ constexpr int p10(int a, int b) { return (5 * a + 2 * b) % 10; }
template <int DIR, int TOT, int OFF, int N2>
__device__ __forceinline__ void d10_s1(cpx (&x)[TOT]) {
    if constexpr (N2 < 5) {
        const cpx a = x[OFF + p10(0, N2)], b = x[OFF + p10(1, N2)];
        x[OFF + p10(0, N2)] = a + b;
        x[OFF + p10(1, N2)] = a - b;
        d10_s1<DIR, TOT, OFF, N2 + 1>(x);
    }
}
template <int DIR, int TOT, int OFF, int K1>
__device__ __forceinline__ void d10_s2(cpx (&x)[TOT]) {
    if constexpr (K1 < 2) {
        r5<DIR>(x[OFF + p10(K1, 0)], x[OFF + p10(K1, 1)], x[OFF + p10(K1, 2)], x[OFF + p10(K1, 3)],
                x[OFF + p10(K1, 4)]);
        d10_s2<DIR, TOT, OFF, K1 + 1>(x);
    }
}
template <int DIR, int TOT, int OFF>
__device__ __forceinline__ void dft10(cpx (&x)[TOT]) {
#ifdef ABL_NOFFT
    return;
#endif
    d10_s1<DIR, TOT, OFF, 0>(x);
    d10_s2<DIR, TOT, OFF, 0>(x);
    cpx t[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) t[k] = x[OFF + p10(k % 2, k % 5)];
#pragma unroll
    for (int k = 0; k < 10; ++k) x[OFF + k] = t[k];
}

}  // namespace ofxfft
