// ofx_synth.hip -- device-side synthetic event generator for bench.py and the
// full-size GPU property tests (v = A * roll(template, d)
// + noise; counter-based so any shard of a run is reproducible from
// (seed, global trace index)).  Not part of the hot path.
#include "ofx_common.h"

struct Philox {
    uint32_t c[4];
    uint32_t k[2];
};

__device__ __forceinline__ void philox_round(uint32_t* c, const uint32_t* k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0];
    const uint32_t n1 = lo1;
    const uint32_t n2 = hi0 ^ c[3] ^ k[1];
    const uint32_t n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t* out) {
    uint32_t c[4] = {c0, c1, c2, c3};
    uint32_t k[2] = {k0, k1};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float u01(uint32_t x) {      // (0,1]
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// White noise + pulse, the HBM-write-bound source of the streaming form of the bench (BASELINE
// configs[4] rehearsal, detprocess_amd.dist.run_sharded): one 128 KiB write per trace and ~15 VALU
// instructions per sample -- a 32-bit counter hash instead of Philox, Box-Muller on the hardware
// log2 / sqrt / sin / cos (1 ulp-ish: a bench source, not a statistics library), so that it can share
// the chip with the fused kernel without becoming the slower of the two (the coloured generator below
// moves ~5x the trace bytes and runs at 1.9 M traces/s; it stays the source of the tests).
// Counter-based: sample n of trace gid depends on (seed, gid, n) only.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {      // "lowbias32" (Wellons)
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float2 box_muller_fast(uint32_t ur, uint32_t ua) {
    // radius from a 24-bit uniform in (0, 1], angle (in turns) from the top bits of ua (16 used)
    const float u = ((float)(ur >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(u));   // sqrt(-2 ln u)
    const float t = (float)(ua >> 8) * (1.0f / 16777216.0f);
    return make_float2(r * __builtin_amdgcn_cosf(t), r * __builtin_amdgcn_sinf(t));
}

// one block per trace, 256 threads, 4 samples per thread per iteration
__global__ __launch_bounds__(256) void k_synth(float* __restrict__ traces,
                                               float* __restrict__ truth, long long first,
                                               int N, const float* __restrict__ tmpl,
                                               float sigma, float amp_lo, float amp_hi,
                                               float pulse_fraction, int max_delay,
                                               unsigned long long seed) {
    const long long b = blockIdx.x;
    const unsigned long long gid = (unsigned long long)(first + b);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t r[4];
    philox4x32((uint32_t)gid, (uint32_t)(gid >> 32), 0xFFFFFFFFu, 0u, k0, k1, r);
    float amp = 0.0f;
    if (u01(r[0]) <= pulse_fraction)
        amp = amp_lo * expf(u01(r[1]) * logf(amp_hi / amp_lo));
    int delay = 0;
    if (max_delay > 0) delay = (int)(r[2] % (uint32_t)(2 * max_delay + 1)) - max_delay;
    if (truth && threadIdx.x == 0) {
        truth[2 * b] = amp;
        truth[2 * b + 1] = (float)delay;
    }
    const uint32_t key = r[3];                       // per-trace key of the sample hash (uniform)
    float* t = traces + (size_t)b * N;
    for (int n4 = threadIdx.x; n4 < N / 4; n4 += 256) {
        const uint32_t c = key + 3u * (uint32_t)n4;
        const uint32_t h0 = hash32(c), h1 = hash32(c + 1u), h2 = hash32(c + 2u);
        const float2 za = box_muller_fast(h0, h2 << 16), zb = box_muller_fast(h1, h2 & 0xffff0000u);
        const float z[4] = {za.x, za.y, zb.x, zb.y};
        float4 v;
        float* pv = &v.x;
        int src = 4 * n4 - delay;
        src = src < 0 ? src + N : src;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int sj = src + j;
            sj = sj >= N ? sj - N : sj;
            pv[j] = fmaf(amp, tmpl[sj], sigma * z[j]);
        }
        *reinterpret_cast<float4*>(t + 4 * n4) = v;
    }
}

// ---------------------------------------------------------------------------
// Coloured noise: noise = irfft(sqrt(J N fs / 2) (xi1 + i xi2)).
// k_spectrum draws the one-sided spectrum, rocFFT C2R (unnormalised) turns it into
// the trace, k_add_pulse adds A * roll(template, d).  noise_amp[k] already carries
// the 1/N of NumPy's irfft.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spectrum(float2* __restrict__ spec, long long first,
                                                  int K, const float* __restrict__ noise_amp,
                                                  unsigned long long seed) {
    const long long b = blockIdx.x;
    const unsigned long long gid = (unsigned long long)(first + b);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    float2* sp = spec + (size_t)b * K;
    for (int kk = threadIdx.x; 2 * kk < K; kk += 256) {
        uint32_t r[4];
        philox4x32((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)kk, 2u, k0, k1, r);
        const float r0 = sqrtf(-2.0f * logf(u01(r[0])));
        float s, c;
        sincospif(2.0f * u01(r[1]), &s, &c);
        const float r1 = sqrtf(-2.0f * logf(u01(r[2])));
        float s2, c2;
        sincospif(2.0f * u01(r[3]), &s2, &c2);
        const int ka = 2 * kk, kb = 2 * kk + 1;
        float2 va = make_float2(r0 * c, r0 * s), vb = make_float2(r1 * c2, r1 * s2);
        // DC and Nyquist are real with variance 2 (synth.coloured_noise)
        if (ka == 0 || ka == K - 1) va = make_float2(va.x * 1.41421356f, 0.0f);
        if (kb == K - 1) vb = make_float2(vb.x * 1.41421356f, 0.0f);
        sp[ka] = make_float2(va.x * noise_amp[ka], va.y * noise_amp[ka]);
        if (kb < K) sp[kb] = make_float2(vb.x * noise_amp[kb], vb.y * noise_amp[kb]);
    }
}

__global__ __launch_bounds__(256) void k_add_pulse(float* __restrict__ traces,
                                                   float* __restrict__ truth, long long first,
                                                   int N, const float* __restrict__ tmpl,
                                                   float amp_lo, float amp_hi,
                                                   float pulse_fraction, int max_delay,
                                                   unsigned long long seed) {
    const long long b = blockIdx.x;
    const unsigned long long gid = (unsigned long long)(first + b);
    uint32_t r[4];
    philox4x32((uint32_t)gid, (uint32_t)(gid >> 32), 0xFFFFFFFFu, 0u, (uint32_t)seed,
               (uint32_t)(seed >> 32), r);
    float amp = 0.0f;
    if (u01(r[0]) <= pulse_fraction) amp = amp_lo * expf(u01(r[1]) * logf(amp_hi / amp_lo));
    int delay = 0;
    if (max_delay > 0) delay = (int)(r[2] % (uint32_t)(2 * max_delay + 1)) - max_delay;
    if (truth && threadIdx.x == 0) {
        truth[2 * b] = amp;
        truth[2 * b + 1] = (float)delay;
    }
    if (amp == 0.0f) return;
    float* t = traces + (size_t)b * N;
    for (int n = threadIdx.x; n < N; n += 256) {
        int src = n - delay;
        if (src < 0) src += N;
        if (src >= N) src -= N;
        t[n] = fmaf(amp, tmpl[src], t[n]);
    }
}

struct SynthFft {
    int n = 0, batch = 0;
    rocfft_plan c2r = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    float2* spec = nullptr;
};
static SynthFft g_sfft;

extern "C" int ofx_synth_release(void) {
    if (g_sfft.c2r) rocfft_plan_destroy(g_sfft.c2r);
    if (g_sfft.info) rocfft_execution_info_destroy(g_sfft.info);
    if (g_sfft.work) (void)hipFree(g_sfft.work);
    if (g_sfft.spec) (void)hipFree(g_sfft.spec);
    g_sfft = SynthFft();
    return OFX_OK;
}

extern "C" int ofx_synth_traces_psd(float* traces, float* truth, long long n_traces,
                                    long long first_index, int n_samples,
                                    const float* template_td, const float* noise_amp,
                                    float amp_lo, float amp_hi, float pulse_fraction,
                                    int max_delay, unsigned long long seed, void* stream) {
    if (!traces || !template_td || !noise_amp || n_traces < 0 || n_samples < 8 ||
        (n_samples & 1) || !(amp_lo > 0) || !(amp_hi >= amp_lo) || max_delay < 0 ||
        max_delay >= n_samples) {
        ofx_set_error("ofx_synth_traces_psd: bad argument");
        return OFX_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int K = n_samples / 2 + 1;
    const int CH = 2048;                                 // traces per FFT batch
    if (g_sfft.n != n_samples) {
        ofx_synth_release();
        {
            const int rc_setup = ofx_rocfft_setup_once();
            if (rc_setup) return rc_setup;
        }
        size_t len = (size_t)n_samples;
        OFX_FFT(rocfft_plan_create(&g_sfft.c2r, rocfft_placement_notinplace,
                                   rocfft_transform_type_real_inverse, rocfft_precision_single,
                                   1, &len, (size_t)CH, nullptr));
        size_t wb = 0;
        OFX_FFT(rocfft_plan_get_work_buffer_size(g_sfft.c2r, &wb));
        OFX_FFT(rocfft_execution_info_create(&g_sfft.info));
        if (wb) {
            OFX_HIP(hipMalloc(&g_sfft.work, wb));
            OFX_FFT(rocfft_execution_info_set_work_buffer(g_sfft.info, g_sfft.work, wb));
        }
        OFX_HIP(hipMalloc(&g_sfft.spec, sizeof(float2) * (size_t)CH * K));
        g_sfft.n = n_samples;
        g_sfft.batch = CH;
    }
    OFX_FFT(rocfft_execution_info_set_stream(g_sfft.info, st));
    // the C2R plan has a fixed batch: a partial last chunk goes through a bounce buffer
    float* bounce = nullptr;
    for (long long b0 = 0; b0 < n_traces; b0 += CH) {
        const int nb = (int)((n_traces - b0 < CH) ? (n_traces - b0) : CH);
        hipLaunchKernelGGL(k_spectrum, dim3(nb), dim3(256), 0, st, g_sfft.spec, first_index + b0,
                           K, noise_amp, seed);
        float* dst = traces + (size_t)b0 * n_samples;
        if (nb < CH) {
            OFX_HIP(hipMalloc(&bounce, sizeof(float) * (size_t)CH * n_samples));
            dst = bounce;
        }
        void* in[1] = {(void*)g_sfft.spec};
        void* out[1] = {(void*)dst};
        OFX_FFT(rocfft_execute(g_sfft.c2r, in, out, g_sfft.info));
        if (nb < CH)
            OFX_HIP(hipMemcpyAsync(traces + (size_t)b0 * n_samples, bounce,
                                   sizeof(float) * (size_t)nb * n_samples,
                                   hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_add_pulse, dim3(nb), dim3(256), 0, st,
                           traces + (size_t)b0 * n_samples, truth ? truth + 2 * b0 : nullptr,
                           first_index + b0, n_samples, template_td, amp_lo, amp_hi,
                           pulse_fraction, max_delay, seed);
    }
    OFX_HIP(hipGetLastError());
    if (bounce) {
        OFX_HIP(hipStreamSynchronize(st));
        (void)hipFree(bounce);
    }
    return OFX_OK;
}

extern "C" int ofx_synth_traces(float* traces, float* truth, long long n_traces,
                                long long first_index, int n_samples,
                                const float* template_td, float sigma, float amp_lo,
                                float amp_hi, float pulse_fraction, int max_delay,
                                unsigned long long seed, void* stream) {
    if (!traces || !template_td || n_traces < 0 || n_samples < 4 || (n_samples & 3) ||
        !(amp_lo > 0) || !(amp_hi >= amp_lo) || max_delay < 0 || max_delay >= n_samples) {
        ofx_set_error("ofx_synth_traces: bad argument");
        return OFX_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long maxgrid = 1 << 30;
    for (long long b0 = 0; b0 < n_traces; b0 += maxgrid) {
        const long long nb = (n_traces - b0 < maxgrid) ? (n_traces - b0) : maxgrid;
        hipLaunchKernelGGL(k_synth, dim3((unsigned)nb), dim3(256), 0, st,
                           traces + (size_t)b0 * n_samples,
                           truth ? truth + 2 * b0 : nullptr, first_index + b0, n_samples,
                           template_td, sigma, amp_lo, amp_hi, pulse_fraction, max_delay,
                           seed);
    }
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
