// ofx_device.h -- device helpers shared by the ROCFFT and FUSED engines:
// wave/block reductions, the arg-max candidate ordering, the low-frequency
// chi2 residual sum and the output-record writer.
#pragma once

#include <hip/hip_runtime.h>

#include "ofx_common.h"

#define OFX_WAVE 64

// Wave64 reductions on the VALU with DPP (no LDS crossbar round trips): xor-1 / xor-2
// inside quads, half-row mirror (8), row mirror (16), row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3; lane 63 then holds the result, which is broadcast as a
// scalar.  The result is valid (and identical) in every lane.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float ofx_dpp(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src),
                                                      CTRL, ROW_MASK, 0xf, BOUND));
}
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ int ofx_dppi(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, BOUND);
}
#define OFX_DPP_STEPS(X)                                                              \
    X(0xB1, 0xf) X(0x4E, 0xf) X(0x141, 0xf) X(0x140, 0xf) X(0x142, 0xa) X(0x143, 0xc)


// ------------------------------------------------------------- candidates
// A candidate of the delay fit: key = A^2, rolled index i, amplitude A.
// Ordering restates NumPy argmin(chi2) on the rolled array: larger A^2 wins
// (chi2 = chi2_0 - A^2 norm), ties -> smaller rolled index.
struct OfxCand {
    float key;
    int idx;
    float amp;
};

__device__ __forceinline__ OfxCand ofx_cand_none() {
    OfxCand c;
    c.key = -1.0f;
    c.idx = 0x7fffffff;
    c.amp = 0.0f;
    return c;
}

__device__ __forceinline__ bool ofx_cand_better(float key, int idx, const OfxCand& b) {
    return (key > b.key) || (key == b.key && idx < b.idx);
}

__device__ __forceinline__ void ofx_cand_take(OfxCand& best, float amp, int idx) {
    const float key = amp * amp;
    if (ofx_cand_better(key, idx, best)) {
        best.key = key;
        best.idx = idx;
        best.amp = amp;
    }
}

// Wave reduce of a candidate (DPP, result broadcast from lane 63 to every lane).
__device__ __forceinline__ OfxCand ofx_cand_wave_reduce(OfxCand c) {
#define STEP(C, M)                                                                    \
    {                                                                                 \
        const float k = ofx_dpp<C, M, false>(c.key, c.key);                           \
        const int i = ofx_dppi<C, M, false>(c.idx, c.idx);                            \
        const float a = ofx_dpp<C, M, false>(c.amp, c.amp);                           \
        if (ofx_cand_better(k, i, c)) {                                               \
            c.key = k;                                                                \
            c.idx = i;                                                                \
            c.amp = a;                                                                \
        }                                                                             \
    }
    OFX_DPP_STEPS(STEP)
#undef STEP
    c.key = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.key), 63));
    c.idx = __builtin_amdgcn_readlane(c.idx, 63);
    c.amp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.amp), 63));
    return c;
}

// Block reduce; result valid in every thread.  scratch: >= (nthreads/64) OfxCand.
// tid: this thread's index (pass an opaque copy to keep the address arithmetic local).
// nwave: waves taking part (default: the whole workgroup; the FUSED kernel passes the waves of
// one half -- the barrier is still the workgroup's).
__device__ __forceinline__ OfxCand ofx_cand_block_reduce(OfxCand c, OfxCand* scratch,
                                                         int tid = -1, int nwave = 0) {
    if (tid < 0) tid = threadIdx.x;
    const int lane = tid & (OFX_WAVE - 1);
    const int wave = tid / OFX_WAVE;
    if (nwave <= 0) nwave = (blockDim.x + OFX_WAVE - 1) / OFX_WAVE;
    c = ofx_cand_wave_reduce(c);
    __syncthreads();
    if (lane == 0) scratch[wave] = c;
    __syncthreads();
    OfxCand r = scratch[0];
    for (int w = 1; w < nwave; ++w) {
        const OfxCand o = scratch[w];
        if (ofx_cand_better(o.key, o.idx, r)) r = o;
    }
    return r;
}

__device__ __forceinline__ float ofx_wave_sum(float v) {
#define STEP(C, M) v += ofx_dpp<C, M, true>(0.0f, v);
    OFX_DPP_STEPS(STEP)
#undef STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float ofx_wave_max(float v) {
#define STEP(C, M) v = fmaxf(v, ofx_dpp<C, M, false>(v, v));
    OFX_DPP_STEPS(STEP)
#undef STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float ofx_wave_min(float v) {
#define STEP(C, M) v = fminf(v, ofx_dpp<C, M, false>(v, v));
    OFX_DPP_STEPS(STEP)
#undef STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Block sum; result valid in every thread.  scratch: >= nthreads/64 floats.
__device__ __forceinline__ float ofx_block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & (OFX_WAVE - 1);
    const int wave = threadIdx.x / OFX_WAVE;
    const int nwave = (blockDim.x + OFX_WAVE - 1) / OFX_WAVE;
    v = ofx_wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.0f;
    for (int w = 0; w < nwave; ++w) r += scratch[w];
    return r;
}
__device__ __forceinline__ float ofx_block_max(float v, float* scratch) {
    const int lane = threadIdx.x & (OFX_WAVE - 1);
    const int wave = threadIdx.x / OFX_WAVE;
    const int nwave = (blockDim.x + OFX_WAVE - 1) / OFX_WAVE;
    v = ofx_wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nwave; ++w) r = fmaxf(r, scratch[w]);
    return r;
}
__device__ __forceinline__ float ofx_block_min(float v, float* scratch) {
    const int lane = threadIdx.x & (OFX_WAVE - 1);
    const int wave = threadIdx.x / OFX_WAVE;
    const int nwave = (blockDim.x + OFX_WAVE - 1) / OFX_WAVE;
    v = ofx_wave_min(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nwave; ++w) r = fminf(r, scratch[w]);
    return r;
}

// ------------------------------------------------------------ low-freq chi2
// One term of  sum_{k < nlow} w_k g_k |V_k - A e^{-2 pi i k (d + frac) / N} S_k|^2 ,
// d = rolled index - pretrigger (integer lag), frac = sub-sample refinement (0 unless the
// search interpolates), w_k = 1 at DC / Nyquist else 2 (one-sided evaluation of the
// two-sided sum; J symmetric, V and S Hermitian).  The integer part of the phase is
// reduced exactly (k d mod N) before it becomes a float.
__device__ __forceinline__ float ofx_lowchi2_term(int k, int N, int d, float amp,
                                                  float2 V, float2 S, float g,
                                                  float frac = 0.0f) {
    int m;
    if ((N & (N - 1)) == 0) {
        m = (int)(((unsigned)k * (unsigned)d) & (unsigned)(N - 1));   // wraps mod 2^32: exact
    } else {
        long long mm = ((long long)k * (long long)d) % (long long)N;
        if (mm < 0) mm += N;
        m = (int)mm;
    }
    float sn, cs;
    sincospif(-2.0f * ((float)m + (float)k * frac) / (float)N, &sn, &cs);
    // ph * S
    const float pr = cs * S.x - sn * S.y;
    const float pi = cs * S.y + sn * S.x;
    const float rr = V.x - amp * pr;
    const float ri = V.y - amp * pi;
    const float w = (k == 0 || 2 * k == N) ? 1.0f : 2.0f;
    return w * g * (rr * rr + ri * ri);
}

// interpolate=True: vertex of the parabola through chi2 at the rolled bins idx-1, idx,
// idx+1 (chi2 = chi0 - A^2 norm) and the amplitude parabola evaluated at the same
// offset (oracle/of1x1.py interpolate_of).  am / a0 / ap: amplitudes at idx-1 / idx /
// idx+1.  No refinement at the array ends, when the three points are not convex, or when the
// vertex lies more than one bin away (window edge with the true minimum outside).
struct OfxRefined {
    float amp, frac, chi2;
};
__device__ __forceinline__ OfxRefined ofx_interpolate(float am, float a0, float ap, int idx,
                                                      int N, float norm, float chi0) {
    OfxRefined r;
    r.amp = a0;
    r.frac = 0.0f;
    r.chi2 = fmaf(-a0 * a0, norm, chi0);
    if (idx <= 0 || idx >= N - 1) return r;
    // y0 - y2 = norm (ap^2 - am^2);  y0 - 2 y1 + y2 = norm (2 a0^2 - am^2 - ap^2)
    const float dpm = (ap - am) * (ap + am);
    const float den = (a0 - am) * (a0 + am) + (a0 - ap) * (a0 + ap);
    if (!(den * norm > 0.0f)) return r;
    const float x = 0.5f * dpm / den;
    if (!(fabsf(x) <= 1.0f)) return r;      // vertex beyond the neighbours: an extrapolation
    r.frac = x;
    r.chi2 = r.chi2 - 0.125f * norm * dpm * dpm / den;
    r.amp = a0 + 0.5f * (ap - am) * x + 0.5f * ((am - a0) + (ap - a0)) * x * x;
    return r;
}

// ----------------------------------------------------------- record writer
__device__ __forceinline__ void ofx_write_search(float* row, const OfxSearchDev& q,
                                                 const OfxSlotDev& sd, float inv_fs,
                                                 int pre, float chi0, OfxCand best,
                                                 float lowchi2, const OfxRefined* ref = nullptr) {
    float* o = row + q.out_off;
    if (best.idx == 0x7fffffff) {
        // no lag compared greater than "none": the trace holds a NaN.  NumPy's argmin
        // returns NaN features for such an event; a finite-looking row would hide it.
        const float qnan = __int_as_float(0x7fc00000);
#pragma unroll
        for (int c = 0; c < OFX_SEARCH_FLOATS; ++c) o[c] = qnan;
        o[OFX_COL_AMPRES] = sd.ampres;
        return;
    }
    const float amp = ref ? ref->amp : best.amp;
    o[OFX_COL_AMP] = amp;
    o[OFX_COL_T0] = ((float)(best.idx - pre) + (ref ? ref->frac : 0.0f)) * inv_fs;
    o[OFX_COL_CHI2] = ref ? ref->chi2 : fmaf(-amp * amp, sd.norm, chi0);
    o[OFX_COL_LOWCHI2] = lowchi2;
    o[OFX_COL_CHI2NOPULSE] = chi0;
    o[OFX_COL_AMPRES] = sd.ampres;
    o[OFX_COL_TIMERES] = 1.0f / sqrtf(amp * amp * sd.tres_sum);
    o[OFX_COL_INDEX] = (float)best.idx;
}
