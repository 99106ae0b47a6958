// ofx_wave_parts.h -- what the one-wave-per-trace kernels share at file scope (ofx_wave.hip: 4096 samples, one
// wave per trace; ofx_wave2.hip: 8192 samples, two waves per trace): the 2048-point geometry of a wave
// (16 x 8 x 16, 64 lanes x 32 complex registers), its LDS record, the stage-1 twiddles from register
// anchors and the pairwise middle step with its rows requested ahead.  Included inside each file's anonymous
// namespace, after ofx_fft_regs.h.
#pragma once

constexpr int WM = 2048;            // complex points of a wave's transform
constexpr int WNV = 32;             // complex values per lane
constexpr int WLD2 = 17;            // D2 row stride (elements)
constexpr int WXB = 128 * WLD2;     // exchange buffer, complex elements (>= 2048)
constexpr int WLOW = 256;           // low bins 2 X_k a wave keeps in LDS
constexpr int WROWS = 256;          // samples per register row n1

struct WaveLds {                    // one per wave
    cpx xb[WXB];                    // exchange buffer / lag dump (4096 floats)
    cpx xlow[WLOW + 8];             // 2 X_k of the wave's low bins
    cpx perm[WNV];                  // lane 0's permutation bounce buffer
};

#include "ofx_fused_parts.h"
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(a, fminf(b, c)); }

// lane 0's two self-paired blocks (0 and 64) in the generic slot shape: the permutation of
// ofx_fused.hip (blocks 0 and 512 there), genA = [A0[0..7], B0[0..7]], genB = [B0[8..15], A0[9..15], A0[0]]
__device__ constexpr int wperm_in_src(int j) {
    if (j < 8) return j;
    if (j < 24) return j + 8;
    return (j - 16 + 1) % 16;
}
__device__ constexpr int wperm_out_src(int j) {
    if (j < 8) return j;
    if (j < 16) return 16 + j - 1;
    return j - 8;
}

// ---- stage-1 twiddles w_2048^{n' k1}: six anchors per lane (k1 = 1, 2, 3 and 4, 8, 12 at n' = lane)
// live in registers for the whole persistent loop; k1 = a + 4 b is the product of two of them, and the
// lane's second virtual thread (n' = lane + 64) differs by the constant w_32^{k1}.  No table loads
// inside the loop: the compiler serialises such loads (load, wait, use, load ...), 30 round trips to
// L2 per stage in the first version of this kernel (profiles/r03_phase_timeline_4096_first_version.json).
// (the anchors pass through an opaque copy per stage: the nine products are recomputed where they are
// used instead of being hoisted out of the persistent loop into 18 more live registers)
__device__ __forceinline__ void t1_opaque(cpx (&A)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(A[i]));
}
template <int K1, bool INV>
__device__ __forceinline__ void t1_step(cpx (&d)[WNV], const cpx (&A)[6]) {
    if constexpr (K1 < 16) {
        constexpr int a = K1 & 3, b = K1 >> 2;
        cpx w;
        if constexpr (b == 0) w = A[a - 1];
        else if constexpr (a == 0) w = A[2 + b];
        else w = cmul(A[a - 1], A[2 + b]);
        if constexpr (!INV) {
            d[K1] = cmul(d[K1], w);
            d[16 + K1] = twmul<K1, -1>(cmul(d[16 + K1], w));
        } else {
            d[K1] = cmulc(d[K1], w);
            d[16 + K1] = twmul<K1, +1>(cmulc(d[16 + K1], w));
        }
        t1_step<K1 + 1, INV>(d, A);
    }
}

// ---- the middle step: 16 pair slots per lane, the filter rows requested WMID_DEPTH slots ahead
#ifndef OFX_WMID_DEPTH
#define OFX_WMID_DEPTH 8
#endif
constexpr int WMID = OFX_WMID_DEPTH;
template <int J>
__device__ __forceinline__ void wmid_request(float4 (&tw)[WMID], cpx (&tg)[WMID], __amdgpu_buffer_rsrc_t rw,
                                             __amdgpu_buffer_rsrc_t rg, int v) {
    tw[J % WMID] = buf_ld4(rw, v * 16, J * 64 * 16);
    tg[J % WMID] = buf_ld2(rg, v * 8, J * 64 * 8);
}
template <int J>
__device__ __forceinline__ void wmid_request_first(float4 (&tw)[WMID], cpx (&tg)[WMID], __amdgpu_buffer_rsrc_t rw,
                                                   __amdgpu_buffer_rsrc_t rg, int v) {
    if constexpr (J < WMID) {
        wmid_request<J>(tw, tg, rw, rg, v);
        wmid_request_first<J + 1>(tw, tg, rw, rg, v);
    }
}
// ODD = 0: the wave holds the bins q = v + 128 j of a transform whose pairs are (q, M - q) -- k_wave, and the
// even wave of k_wave2 --, lane 0 with the two self-paired blocks; ODD = 1: pairs (q, M - 1 - q) -- the odd wave of
// k_wave2 (global bins 2 q + 1) --, no self-paired block.  The two differ in where the low bins land.
// ODD = 2: a crossing wave of k_wave2 with four waves (its partners are another wave's bins): the partner side of
// the low bins goes to that wave's stash, LP.
template <int J, int ODD = 0>
__device__ __forceinline__ void wmid(cpx (&d)[WNV], __amdgpu_buffer_rsrc_t rw, __amdgpu_buffer_rsrc_t rg,
                                     int v, WaveLds& L, cpx tlo, cpx thi, float4 (&tw)[WMID],
                                     cpx (&tg)[WMID], cpx& chi, [[maybe_unused]] WaveLds* LP = nullptr) {
    if constexpr (J < 16) {
        const float4 w = tw[J % WMID];
        const cpx g = tg[J % WMID];
        if constexpr (J + WMID < 16) wmid_request<J + WMID>(tw, tg, rw, rg, v);
        const cpx T = twmul<J, -1>(J < 8 ? tlo : thi);
        cpx xk2, xp2;
        mid_slot(d[J], d[16 + 15 - J], T, w, g, xk2, xp2, chi);
        // low bins for lowchi2 / psd_amp: xk2 = 2 X_k, k = v + 128 J; xp2 = 2 conj(X_p),
        // p = 128 (16 - J) - v (v != 0); lane 0's slots 8, 9 hold the bins 64 and 192.  No branch:
        // what does not apply goes to the padding behind the stash.
        if constexpr (J <= 1) L.xlow[v + 128 * J] = xk2;
        if constexpr (ODD == 0) {
            if constexpr (J >= 14) L.xlow[v != 0 ? 128 * (16 - J) - v : WLOW + 1] = cconj(xp2);
            if constexpr (J == 8 || J == 9) L.xlow[v == 0 ? 64 + 128 * (J - 8) : WLOW + 2] = xk2;
        } else if constexpr (ODD == 1) {
            if constexpr (J >= 14) L.xlow[128 * (16 - J) - 1 - v] = cconj(xp2);     // p = 127 - v + 128 (15 - J)
        } else {
            if constexpr (J >= 14) LP->xlow[128 * (16 - J) - 1 - v] = cconj(xp2);
        }
        wmid<J + 1, ODD>(d, rw, rg, v, L, tlo, thi, tw, tg, chi, LP);
    }
}
