// ofx_ingest.hip -- cut-and-convert front end of ofx_process_adc (include/ofx.h):
// events are windows [trigger - n_pretrigger, + n_samples) of continuous int16 ADC streams
// (processing_data.py:640-656: min_idx = trigger_index - nb_pretrigger_samples, max_idx =
// min_idx + nb_samples, cut only if 0 <= min_idx and max_idx <= stream length), converted
// to amps by a per-channel affine map.  HBM-bound byte work: 2 B read + 4 B written per
// sample, 8 samples per thread (one 16-byte store pair), streams read through L2 so that
// overlapping windows of neighbouring triggers are fetched from HBM once.
#include <hip/hip_runtime.h>

#include "ofx_common.h"

namespace {

constexpr int CUT_THREADS = 256;
constexpr int CUT_PER_THREAD = 8;
constexpr int CUT_MAX_CH = 32;

struct CutCoef {
    float scale[CUT_MAX_CH];
    float offset[CUT_MAX_CH];
};

__global__ __launch_bounds__(CUT_THREADS) void k_cut(const int16_t* __restrict__ adc,
                                                     long long n_stream, int n_channels, int N,
                                                     int pre, const long long* __restrict__ trig,
                                                     CutCoef cf, float* __restrict__ events,
                                                     uint8_t* __restrict__ valid) {
    const long long b = blockIdx.y;
    const int c = blockIdx.z;
    const long long lo = trig[b] - pre;
    const bool ok = (lo >= 0) && (lo + N <= n_stream);
    if (blockIdx.x == 0 && c == 0 && threadIdx.x == 0) valid[b] = ok ? 1 : 0;
    float* dst = events + ((size_t)b * n_channels + c) * N;
    const int j0 = (blockIdx.x * CUT_THREADS + threadIdx.x) * CUT_PER_THREAD;
    if (j0 >= N) return;
    const float sc = cf.scale[c], of = cf.offset[c];
    const int16_t* src = adc + (size_t)c * n_stream + lo + j0;
    float v[CUT_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CUT_PER_THREAD; ++i) {
        const bool in = ok && (j0 + i < N);
        const float a = in ? (float)src[i] : 0.0f;
        // product and sum rounded separately, as NumPy float32 arithmetic does: the asm
        // keeps the compiler from contracting them into one FMA
        float pr = a * sc;
        asm volatile("" : "+v"(pr));
        v[i] = pr + of;
    }
    if (j0 + CUT_PER_THREAD <= N && (N % 4) == 0) {
        float4* d4 = reinterpret_cast<float4*>(dst + j0);
        d4[0] = make_float4(v[0], v[1], v[2], v[3]);
        d4[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int i = 0; i < CUT_PER_THREAD; ++i)
            if (j0 + i < N) dst[j0 + i] = v[i];
    }
}

}  // namespace

int ofx_cut_launch(const int16_t* d_adc, long long n_stream, int n_channels, int n_samples,
                   int n_pretrigger, const long long* d_trig, long long nb, const float* scale,
                   const float* offset, float* d_events, uint8_t* d_valid, hipStream_t st) {
    if (n_channels > CUT_MAX_CH) {
        ofx_set_error("ofx_process_adc: more than %d channels", CUT_MAX_CH);
        return OFX_ERR_UNSUPPORTED;
    }
    if (nb <= 0) return OFX_OK;
    if (nb > 65535) {
        ofx_set_error("ofx_process_adc: internal chunk of %lld events exceeds the grid limit", nb);
        return OFX_ERR_ARG;
    }
    CutCoef cf;
    for (int c = 0; c < CUT_MAX_CH; ++c) {
        cf.scale[c] = c < n_channels ? scale[c] : 0.0f;
        cf.offset[c] = c < n_channels ? offset[c] : 0.0f;
    }
    const int per_block = CUT_THREADS * CUT_PER_THREAD;
    dim3 grid((unsigned)((n_samples + per_block - 1) / per_block), (unsigned)nb, (unsigned)n_channels);
    hipLaunchKernelGGL(k_cut, grid, dim3(CUT_THREADS), 0, st, d_adc, n_stream, n_channels,
                       n_samples, n_pretrigger, d_trig, cf, d_events, d_valid);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
