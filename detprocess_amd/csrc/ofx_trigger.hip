// ofx_trigger.hip -- continuous-data optimal-filter trigger (include/ofx.h, "ofx_trigger"):
//   update_trace : overlap-save FIR filtering of a long stream with batched rocFFT
//                  (blocks of P samples advanced by H = P - (N - 1), gathered with the
//                  int16 / baseline conversion), filtered = conv / vscale,
//                  delta chi2 = filtered^2 w, edges zeroed   (oftrigger.py:649-679)
//   find         : threshold, range merging with a static pile-up window, arg-max per range
//                  (oftrigger.py:976-1019, :29-77) as two device scans + one atomic-max pass
// All passes are HBM-bound element-wise / scan work around two rocFFT calls.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ofx_common.h"

#define TRIG_MAX 4

struct ofx_trigger {
    int N = 0, pre = 0, device = 0;
    int C = 1, M = 1;                     // channels x amplitudes (1 x 1: the fused fast path)
    double fs = 0, vscale = 1, w = 1;
    double iw[TRIG_MAX * TRIG_MAX] = {0}, wm[TRIG_MAX * TRIG_MAX] = {0};   // N x M matrices
    int P = 0, H = 0;                     // FFT block length, hop
    float2* d_hfft = nullptr;             // [C][M][P/2+1] FFT of the zero-padded filters / P
    float2* d_acc = nullptr;  size_t acc_elems = 0;      // N x M: accumulated spectrum of one amplitude
    float* d_vtd = nullptr;   size_t vtd_elems = 0;      // N x M: [M][n] summed convolutions
    // per-stream state
    long long n = 0, nblk = 0;
    float* d_xpad = nullptr;  size_t xpad_elems = 0;
    float2* d_spec = nullptr; size_t spec_elems = 0;
    float* d_yblk = nullptr;  size_t yblk_elems = 0;
    float* d_filt = nullptr;  float* d_dchi = nullptr; size_t trace_elems = 0;
    int* d_scan = nullptr;    size_t scan_elems = 0;     // last-above / range-start scans
    unsigned long long* d_key = nullptr; size_t key_elems = 0;
    void* d_tmp = nullptr;    size_t tmp_bytes = 0;
    void* d_stage = nullptr;  size_t stage_bytes = 0;
    long long* d_count = nullptr;
    long long* d_oidx = nullptr; float* d_odchi = nullptr; float* d_oamp = nullptr; size_t out_cap = 0;
    rocfft_plan r2c = nullptr, c2r = nullptr;
    rocfft_execution_info info = nullptr;
    void* d_work = nullptr; size_t work_bytes = 0;
    long long plan_nblk = 0;
    // residual pass (oftrigger.py:752-845)
    float* d_dchi_saved = nullptr; size_t saved_elems = 0; bool saved = false;
    float* d_pulse = nullptr;             // [M][M][N] delta-chi2 pulse-shape table G
    long long* d_sel = nullptr; size_t sel_elems = 0;    // compaction / trigger-index staging
    float* d_selv = nullptr;  size_t selv_elems = 0;
};

namespace {

template <typename T>
int grow(T** buf, size_t* have, size_t want) {
    if (*have >= want) return OFX_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    OFX_HIP(hipMalloc(reinterpret_cast<void**>(buf), want * sizeof(T)));
    *have = want;
    return OFX_OK;
}

constexpr int TB = 256;

// block j of the overlap-save scheme, gathered explicitly (rocFFT batches whose input
// distance is smaller than the transform length came back wrong for every batch but the
// first): seg[j][i] = x[j H + i - (N-1)] - c0 inside the stream, -c0 outside.  The filter
// has no DC gain, so removing the constant c0 = x[0] is exact and keeps the fp32 FFT error
// small for streams riding on a large baseline.
template <typename T>
__global__ void k_pad(const T* __restrict__ x, long long n, int front, int P, int H,
                      long long total, float scale, float offset, float* __restrict__ seg) {
    const long long idx = (long long)blockIdx.x * TB + threadIdx.x;
    if (idx >= total) return;
    const float c0 = (float)x[0] * scale + offset;
    const long long blk = idx / P;
    const long long j = blk * H + (idx - blk * P) - front;
    float v = 0.0f;
    if (j >= 0 && j < n) v = (float)x[j] * scale + offset;
    seg[idx] = v - c0;
}

__global__ void k_mul(float2* __restrict__ spec, const float2* __restrict__ h, int K,
                      long long total) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= total) return;
    const float2 a = spec[i], b = h[i % K];
    spec[i] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// 'same' output t corresponds to full-convolution sample nf = t + (N-1)/2, which is sample
// (N-1) + nf % H of block nf / H.  The constant removed in k_pad does not come back (no DC).
__global__ void k_dchi2(const float* __restrict__ yblk, long long n, int N, int P, int H,
                        float inv_vscale, float w, int padding, float* __restrict__ filt,
                        float* __restrict__ dchi) {
    const long long t = (long long)blockIdx.x * TB + threadIdx.x;
    if (t >= n) return;
    const long long nf = t + (N - 1) / 2;
    const long long blk = nf / H;
    const int off = (int)(nf - blk * H) + (N - 1);
    const float a = yblk[blk * P + off] * inv_vscale;
    filt[t] = a;
    float d = a * w * a;
    if (padding) {
        // oftrigger.py:676-679: [:N] = 0 and [-(N) + (N+1)%2:] = 0
        const long long tail = (long long)N - ((N + 1) % 2);
        if (t < N || t >= n - tail) d = 0.0f;
    }
    dchi[t] = d;
}

struct TrigMat {
    float iw[TRIG_MAX * TRIG_MAX];
    float w[TRIG_MAX * TRIG_MAX];
};

// N x M: spectrum of amplitude m = sum over channels of spec_b x h[b][m]  (oftrigger.py:656-662:
// the per-channel convolutions are summed)
__global__ void k_mulacc(const float2* __restrict__ spec, const float2* __restrict__ h, int C,
                         int M, int m, int K, long long per_chan, float2* __restrict__ acc) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= per_chan) return;
    const int k = (int)(i % K);
    float2 r = make_float2(0.0f, 0.0f);
    for (int b = 0; b < C; ++b) {
        const float2 a = spec[(size_t)b * per_chan + i], f = h[((size_t)b * M + m) * K + k];
        r.x += a.x * f.x - a.y * f.y;
        r.y += a.x * f.y + a.y * f.x;
    }
    acc[i] = r;
}

// valid part of the overlap-save blocks of one amplitude -> V_td[m][t] ('same' alignment as k_dchi2)
__global__ void k_extract(const float* __restrict__ yblk, long long n, int N, int P, int H,
                          float* __restrict__ vtd) {
    const long long t = (long long)blockIdx.x * TB + threadIdx.x;
    if (t >= n) return;
    const long long nf = t + (N - 1) / 2;
    const long long blk = nf / H;
    vtd[t] = yblk[blk * P + (int)(nf - blk * H) + (N - 1)];
}

// filtered = iw V_td ; delta chi2 = filtered^T w filtered ; edges as in k_dchi2 (oftrigger.py:663-679)
__global__ void k_combine(const float* __restrict__ vtd, long long n, int N, int M, TrigMat mat,
                          int padding, float* __restrict__ filt, float* __restrict__ dchi) {
    const long long t = (long long)blockIdx.x * TB + threadIdx.x;
    if (t >= n) return;
    float v[TRIG_MAX], a[TRIG_MAX];
#pragma unroll
    for (int j = 0; j < TRIG_MAX; ++j) v[j] = j < M ? vtd[(size_t)j * n + t] : 0.0f;
#pragma unroll
    for (int i = 0; i < TRIG_MAX; ++i) {
        float r = 0.0f;
#pragma unroll
        for (int j = 0; j < TRIG_MAX; ++j) r += mat.iw[i * TRIG_MAX + j] * v[j];
        a[i] = r;
        if (i < M) filt[(size_t)i * n + t] = r;
    }
    float d = 0.0f;
#pragma unroll
    for (int i = 0; i < TRIG_MAX; ++i) {
        float r = 0.0f;
#pragma unroll
        for (int j = 0; j < TRIG_MAX; ++j) r += mat.w[i * TRIG_MAX + j] * a[j];
        d += a[i] * r;
    }
    if (padding) {
        const long long tail = (long long)N - ((N + 1) % 2);
        if (t < N || t >= n - tail) d = 0.0f;
    }
    dchi[t] = d;
}

__global__ void k_above(const float* __restrict__ dchi, long long n, float thr,
                        int* __restrict__ last) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    last[i] = (dchi[i] > thr) ? (int)i : -1;
}

// prev[i] = index of the last above-threshold sample <= i (inclusive max scan of k_above).
// A range starts at an above-threshold sample whose predecessor is more than `window` away.
__global__ void k_starts(const float* __restrict__ dchi, const int* __restrict__ prev,
                         long long n, float thr, long long window, int* __restrict__ start) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    int s = -1;
    if (dchi[i] > thr) {
        const int pv = (i > 0) ? prev[i - 1] : -1;
        if (pv < 0 || (i - pv) > window) s = (int)i;
    }
    start[i] = s;
}

// rstart[i] = start of the range sample i belongs to (inclusive max scan of k_starts).
// key = (delta chi2 bits, inverted offset): atomicMax keeps the first maximum (np.argmax).
__global__ void k_best(const float* __restrict__ dchi, const int* __restrict__ rstart,
                       long long n, float thr, unsigned long long* __restrict__ key) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const float d = dchi[i];
    if (!(d > thr)) return;
    const int rs = rstart[i];
    const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) |
                                 (unsigned long long)(0xFFFFFFFFu - (unsigned)(i - rs));
    atomicMax(&key[rs], k);
}

__global__ void k_emit(const float* __restrict__ dchi, const float* __restrict__ filt,
                       const int* __restrict__ prev, const unsigned long long* __restrict__ key,
                       long long n, float thr, long long window, long long cap, int M,
                       long long* __restrict__ count, long long* __restrict__ oidx,
                       float* __restrict__ odchi, float* __restrict__ oamp) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    if (!(dchi[i] > thr)) return;
    const int pv = (i > 0) ? prev[i - 1] : -1;
    if (!(pv < 0 || (i - pv) > window)) return;            // not a range start
    const unsigned long long k = key[i];
    const long long best = i + (long long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
    const long long slot = (long long)atomicAdd(reinterpret_cast<unsigned long long*>(count), 1ull);
    if (slot < cap) {
        oidx[slot] = best;
        odchi[slot] = dchi[best];
        for (int m = 0; m < M; ++m) oamp[slot * M + m] = filt[(size_t)m * n + best];
    }
}

int blocks_for(long long n) { return (int)((n + TB - 1) / TB); }

int make_plans(ofx_trigger* t, long long nblk, hipStream_t st) {
    if (t->plan_nblk == nblk && t->r2c) {
        OFX_FFT(rocfft_execution_info_set_stream(t->info, st));
        return OFX_OK;
    }
    if (t->r2c) rocfft_plan_destroy(t->r2c);
    if (t->c2r) rocfft_plan_destroy(t->c2r);
    if (t->info) rocfft_execution_info_destroy(t->info);
    t->r2c = t->c2r = nullptr;
    t->info = nullptr;
    {
        const int rc_setup = ofx_rocfft_setup_once();
        if (rc_setup) return rc_setup;
    }
    const size_t len = (size_t)t->P;
    OFX_FFT(rocfft_plan_create(&t->r2c, rocfft_placement_notinplace,
                               rocfft_transform_type_real_forward, rocfft_precision_single, 1,
                               &len, (size_t)nblk, nullptr));
    OFX_FFT(rocfft_plan_create(&t->c2r, rocfft_placement_notinplace,
                               rocfft_transform_type_real_inverse, rocfft_precision_single, 1,
                               &len, (size_t)nblk, nullptr));
    size_t w1 = 0, w2 = 0;
    OFX_FFT(rocfft_plan_get_work_buffer_size(t->r2c, &w1));
    OFX_FFT(rocfft_plan_get_work_buffer_size(t->c2r, &w2));
    const size_t wb = std::max(w1, w2);
    if (wb > t->work_bytes) {
        if (t->d_work) (void)hipFree(t->d_work);
        t->d_work = nullptr;
        t->work_bytes = 0;
        OFX_HIP(hipMalloc(&t->d_work, wb));
        t->work_bytes = wb;
    }
    OFX_FFT(rocfft_execution_info_create(&t->info));
    if (wb) OFX_FFT(rocfft_execution_info_set_work_buffer(t->info, t->d_work, t->work_bytes));
    OFX_FFT(rocfft_execution_info_set_stream(t->info, st));
    t->plan_nblk = nblk;
    return OFX_OK;
}

}  // namespace

namespace {

// FFT of one zero-padded filter in fp64 on the host (iterative radix-2, P a power of two),
// scaled by 1/P (rocFFT's inverse is unnormalised); K = P/2 + 1 bins into h
void padded_filter_fft(const double* phi_td, int n_samples, int P, float2* h) {
    std::vector<double> re(P, 0.0), im(P, 0.0);
    for (int i = 0; i < n_samples; ++i) re[i] = phi_td[i];
    for (int i = 1, j = 0; i < P; ++i) {
        int bit = P >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (int len = 2; len <= P; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < P; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = std::cos(ang * k), wi = std::sin(ang * k);
                const double ur = re[i + k], ui = im[i + k];
                const double vr = re[i + k + len / 2] * wr - im[i + k + len / 2] * wi;
                const double vi = re[i + k + len / 2] * wi + im[i + k + len / 2] * wr;
                re[i + k] = ur + vr; im[i + k] = ui + vi;
                re[i + k + len / 2] = ur - vr; im[i + k + len / 2] = ui - vi;
            }
    }
    const int K = P / 2 + 1;
    for (int k = 0; k < K; ++k) h[k] = make_float2((float)(re[k] / P), (float)(im[k] / P));
}

int create_common(ofx_trigger** out, int n_samples, int n_pretrigger, double fs, int C, int M,
                  const double* phi_td, int device) {
    OFX_HIP(hipSetDevice(device));
    ofx_trigger* t = new ofx_trigger();
    t->N = n_samples;
    t->pre = n_pretrigger;
    t->fs = fs;
    t->C = C;
    t->M = M;
    t->device = device;
    int P = 1;
    while (P < 4 * n_samples) P <<= 1;                 // >= 75 % of every block is new output
    if (P < 4096) P = 4096;
    t->P = P;
    t->H = P - (n_samples - 1);
    const int K = P / 2 + 1;
    std::vector<float2> h((size_t)C * M * K);
    for (int r = 0; r < C * M; ++r)
        padded_filter_fft(phi_td + (size_t)r * n_samples, n_samples, P, h.data() + (size_t)r * K);
    if (hipMalloc(&t->d_hfft, sizeof(float2) * h.size()) != hipSuccess ||
        hipMemcpy(t->d_hfft, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice) !=
            hipSuccess ||
        hipMalloc(&t->d_count, sizeof(long long)) != hipSuccess) {
        ofx_set_error("ofx_trigger_create: device allocation failed");
        delete t;
        return OFX_ERR_HIP;
    }
    *out = t;
    return OFX_OK;
}

}  // namespace

extern "C" int ofx_trigger_create(ofx_trigger** out, int n_samples, int n_pretrigger, double fs,
                                  const double* phi_td, double vscale, double w, int device) {
    if (!out || n_samples < 2 || n_pretrigger < 0 || n_pretrigger >= n_samples || !(fs > 0) ||
        !phi_td || !(vscale != 0.0) || !(w > 0)) {
        ofx_set_error("ofx_trigger_create: bad argument");
        return OFX_ERR_ARG;
    }
    int rc = create_common(out, n_samples, n_pretrigger, fs, 1, 1, phi_td, device);
    if (rc) return rc;
    (*out)->vscale = vscale;
    (*out)->w = w;
    (*out)->iw[0] = 1.0 / vscale;
    (*out)->wm[0] = w;
    return OFX_OK;
}

extern "C" int ofx_trigger_create_nxm(ofx_trigger** out, int n_samples, int n_pretrigger,
                                      double fs, int n_chan, int n_amp, const double* phi_td,
                                      const double* iw, const double* w, int device) {
    if (!out || n_samples < 2 || n_pretrigger < 0 || n_pretrigger >= n_samples || !(fs > 0) ||
        n_chan < 1 || n_chan > TRIG_MAX || n_amp < 1 || n_amp > TRIG_MAX || !phi_td || !iw || !w) {
        ofx_set_error("ofx_trigger_create_nxm: bad argument (1..%d channels and amplitudes)",
                      TRIG_MAX);
        return OFX_ERR_ARG;
    }
    int rc = create_common(out, n_samples, n_pretrigger, fs, n_chan, n_amp, phi_td, device);
    if (rc) return rc;
    ofx_trigger* t = *out;
    for (int i = 0; i < n_amp; ++i)
        for (int j = 0; j < n_amp; ++j) {
            t->iw[i * TRIG_MAX + j] = iw[i * n_amp + j];
            t->wm[i * TRIG_MAX + j] = w[i * n_amp + j];
        }
    t->vscale = 1.0 / t->iw[0];
    t->w = t->wm[0];
    return OFX_OK;
}

extern "C" int ofx_trigger_destroy(ofx_trigger* t) {
    if (!t) return OFX_OK;
    (void)hipSetDevice(t->device);
    if (t->r2c) rocfft_plan_destroy(t->r2c);
    if (t->c2r) rocfft_plan_destroy(t->c2r);
    if (t->info) rocfft_execution_info_destroy(t->info);
    void* bufs[] = {t->d_hfft, t->d_xpad, t->d_spec, t->d_yblk, t->d_filt, t->d_dchi, t->d_scan,
                    t->d_key, t->d_tmp, t->d_stage, t->d_count, t->d_oidx, t->d_odchi, t->d_oamp,
                    t->d_work, t->d_acc, t->d_vtd, t->d_dchi_saved, t->d_pulse, t->d_sel,
                    t->d_selv};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete t;
    return OFX_OK;
}

extern "C" int ofx_trigger_update_traces(ofx_trigger* t, const void* x, int dtype, long long n,
                                         int mem, const double* scale, const double* offset,
                                         int padding, void* stream) {
    if (!t || !x || n < 1 || (dtype != 0 && dtype != 1) || n > 2000000000LL ||
        (dtype == 1 && (!scale || !offset))) {
        ofx_set_error("ofx_trigger_update_traces: bad argument");
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    const int N = t->N, P = t->P, H = t->H, K = P / 2 + 1, C = t->C, M = t->M;
    const long long nfull_needed = n + (N - 1) / 2;             // full-conv samples used
    const long long nblk = (nfull_needed + H - 1) / H;
    const long long total = nblk * P;                            // gathered blocks
    const size_t esz = dtype == 0 ? 4 : 2;
    const char* d_x = (const char*)x;
    if (mem == OFX_MEM_HOST) {
        const size_t want = (size_t)C * n * esz;
        if (t->stage_bytes < want) {
            if (t->d_stage) (void)hipFree(t->d_stage);
            t->d_stage = nullptr;
            t->stage_bytes = 0;
            OFX_HIP(hipMalloc(&t->d_stage, want));
            t->stage_bytes = want;
        }
        OFX_HIP(hipMemcpyAsync(t->d_stage, x, want, hipMemcpyHostToDevice, st));
        d_x = (const char*)t->d_stage;
    }
    int rc;
    if ((rc = grow(&t->d_xpad, &t->xpad_elems, (size_t)total))) return rc;
    if ((rc = grow(&t->d_spec, &t->spec_elems, (size_t)C * nblk * K))) return rc;
    if ((rc = grow(&t->d_yblk, &t->yblk_elems, (size_t)nblk * P))) return rc;
    if (t->trace_elems < (size_t)n) {
        if (t->d_filt) (void)hipFree(t->d_filt);
        if (t->d_dchi) (void)hipFree(t->d_dchi);
        t->d_filt = t->d_dchi = nullptr;
        t->trace_elems = 0;
        OFX_HIP(hipMalloc(&t->d_filt, (size_t)M * n * sizeof(float)));
        OFX_HIP(hipMalloc(&t->d_dchi, (size_t)n * sizeof(float)));
        t->trace_elems = (size_t)n;
    }
    if ((rc = make_plans(t, nblk, st))) return rc;
    for (int b = 0; b < C; ++b) {
        const char* xb = d_x + (size_t)b * n * esz;
        if (dtype == 0)
            hipLaunchKernelGGL(k_pad<float>, dim3(blocks_for(total)), dim3(TB), 0, st,
                               (const float*)xb, n, N - 1, P, H, total, 1.0f, 0.0f, t->d_xpad);
        else
            hipLaunchKernelGGL(k_pad<int16_t>, dim3(blocks_for(total)), dim3(TB), 0, st,
                               (const int16_t*)xb, n, N - 1, P, H, total, (float)scale[b],
                               (float)offset[b], t->d_xpad);
        void* in1[1] = {t->d_xpad};
        void* out1[1] = {t->d_spec + (size_t)b * nblk * K};
        OFX_FFT(rocfft_execute(t->r2c, in1, out1, t->info));
    }
    if (C == 1 && M == 1) {
        hipLaunchKernelGGL(k_mul, dim3(blocks_for(nblk * K)), dim3(TB), 0, st, t->d_spec,
                           t->d_hfft, K, nblk * K);
        void* in2[1] = {t->d_spec};
        void* out2[1] = {t->d_yblk};
        OFX_FFT(rocfft_execute(t->c2r, in2, out2, t->info));
        hipLaunchKernelGGL(k_dchi2, dim3(blocks_for(n)), dim3(TB), 0, st, t->d_yblk, n, N, P, H,
                           (float)(1.0 / t->vscale), (float)t->w, padding, t->d_filt, t->d_dchi);
    } else {
        if ((rc = grow(&t->d_acc, &t->acc_elems, (size_t)nblk * K))) return rc;
        if ((rc = grow(&t->d_vtd, &t->vtd_elems, (size_t)M * n))) return rc;
        for (int m = 0; m < M; ++m) {
            hipLaunchKernelGGL(k_mulacc, dim3(blocks_for(nblk * K)), dim3(TB), 0, st, t->d_spec,
                               t->d_hfft, C, M, m, K, nblk * K, t->d_acc);
            void* in2[1] = {t->d_acc};
            void* out2[1] = {t->d_yblk};
            OFX_FFT(rocfft_execute(t->c2r, in2, out2, t->info));
            hipLaunchKernelGGL(k_extract, dim3(blocks_for(n)), dim3(TB), 0, st, t->d_yblk, n, N, P,
                               H, t->d_vtd + (size_t)m * n);
        }
        TrigMat mat;
        for (int i = 0; i < TRIG_MAX * TRIG_MAX; ++i) {
            mat.iw[i] = (float)t->iw[i];
            mat.w[i] = (float)t->wm[i];
        }
        hipLaunchKernelGGL(k_combine, dim3(blocks_for(n)), dim3(TB), 0, st, t->d_vtd, n, N, M, mat,
                           padding, t->d_filt, t->d_dchi);
    }
    OFX_HIP(hipGetLastError());
    t->n = n;
    t->saved = false;
    t->nblk = nblk;
    return OFX_OK;
}

extern "C" int ofx_trigger_update_trace(ofx_trigger* t, const void* x, int dtype, long long n,
                                        int mem, double scale, double offset, int padding,
                                        void* stream) {
    if (t && t->C != 1) {
        ofx_set_error("ofx_trigger_update_trace: %d-channel trigger, use ofx_trigger_update_traces",
                      t->C);
        return OFX_ERR_ARG;
    }
    return ofx_trigger_update_traces(t, x, dtype, n, mem, &scale, &offset, padding, stream);
}

extern "C" int ofx_trigger_get_traces(ofx_trigger* t, float* filtered, float* dchi, int mem,
                                      void* stream) {
    if (!t || t->n == 0) {
        ofx_set_error("ofx_trigger_get_traces: no trace (call ofx_trigger_update_trace first)");
        return OFX_ERR_STATE;
    }
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    const hipMemcpyKind kind = mem == OFX_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (filtered)          // [n_amp][n]
        OFX_HIP(hipMemcpyAsync(filtered, t->d_filt, (size_t)t->M * t->n * 4, kind, st));
    if (dchi) OFX_HIP(hipMemcpyAsync(dchi, t->d_dchi, (size_t)t->n * 4, kind, st));
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

extern "C" int ofx_trigger_find(ofx_trigger* t, double chi2_threshold, long long window,
                                long long* index, float* dchi_out, float* amp_out,
                                long long cap, long long* n_out, void* stream) {
    if (!t || t->n == 0) {
        ofx_set_error("ofx_trigger_find: no trace (call ofx_trigger_update_trace first)");
        return OFX_ERR_STATE;
    }
    if (!n_out || cap < 0 || (cap > 0 && (!index || !dchi_out || !amp_out)) || window < 0) {
        ofx_set_error("ofx_trigger_find: bad argument");
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    const long long n = t->n;
    const float thr = (float)chi2_threshold;
    int rc;
    if ((rc = grow(&t->d_scan, &t->scan_elems, (size_t)2 * n))) return rc;
    if ((rc = grow(&t->d_key, &t->key_elems, (size_t)n))) return rc;
    if (t->out_cap < (size_t)cap) {
        if (t->d_oidx) (void)hipFree(t->d_oidx);
        if (t->d_odchi) (void)hipFree(t->d_odchi);
        if (t->d_oamp) (void)hipFree(t->d_oamp);
        t->d_oidx = nullptr; t->d_odchi = t->d_oamp = nullptr;
        t->out_cap = 0;
        OFX_HIP(hipMalloc(&t->d_oidx, (size_t)cap * sizeof(long long)));
        OFX_HIP(hipMalloc(&t->d_odchi, (size_t)cap * sizeof(float)));
        OFX_HIP(hipMalloc(&t->d_oamp, (size_t)cap * t->M * sizeof(float)));
        t->out_cap = (size_t)cap;
    }
    int* prev = t->d_scan;
    int* rstart = t->d_scan + n;
    size_t need = 0;
    OFX_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, need, prev, prev, hipcub::Max(), (int)n, st));
    if (need > t->tmp_bytes) {
        if (t->d_tmp) (void)hipFree(t->d_tmp);
        t->d_tmp = nullptr;
        t->tmp_bytes = 0;
        OFX_HIP(hipMalloc(&t->d_tmp, need));
        t->tmp_bytes = need;
    }
    const int nb = blocks_for(n);
    hipLaunchKernelGGL(k_above, dim3(nb), dim3(TB), 0, st, t->d_dchi, n, thr, prev);
    size_t tb = t->tmp_bytes;
    OFX_HIP(hipcub::DeviceScan::InclusiveScan(t->d_tmp, tb, prev, prev, hipcub::Max(), (int)n, st));
    hipLaunchKernelGGL(k_starts, dim3(nb), dim3(TB), 0, st, t->d_dchi, prev, n, thr, window, rstart);
    tb = t->tmp_bytes;
    OFX_HIP(hipcub::DeviceScan::InclusiveScan(t->d_tmp, tb, rstart, rstart, hipcub::Max(), (int)n, st));
    OFX_HIP(hipMemsetAsync(t->d_key, 0, (size_t)n * sizeof(unsigned long long), st));
    OFX_HIP(hipMemsetAsync(t->d_count, 0, sizeof(long long), st));
    hipLaunchKernelGGL(k_best, dim3(nb), dim3(TB), 0, st, t->d_dchi, rstart, n, thr, t->d_key);
    hipLaunchKernelGGL(k_emit, dim3(nb), dim3(TB), 0, st, t->d_dchi, t->d_filt, prev, t->d_key, n,
                       thr, window, cap, t->M, t->d_count, t->d_oidx, t->d_odchi, t->d_oamp);
    OFX_HIP(hipGetLastError());
    long long cnt = 0;
    OFX_HIP(hipMemcpyAsync(&cnt, t->d_count, sizeof(long long), hipMemcpyDeviceToHost, st));
    OFX_HIP(hipStreamSynchronize(st));
    *n_out = cnt;
    const long long m = std::min(cnt, cap);
    if (m > 0) {
        std::vector<long long> hi(m);
        const int M = t->M;
        std::vector<float> hd(m), ha((size_t)m * M);
        OFX_HIP(hipMemcpy(hi.data(), t->d_oidx, (size_t)m * sizeof(long long), hipMemcpyDeviceToHost));
        OFX_HIP(hipMemcpy(hd.data(), t->d_odchi, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
        OFX_HIP(hipMemcpy(ha.data(), t->d_oamp, (size_t)m * M * sizeof(float),
                          hipMemcpyDeviceToHost));
        std::vector<long long> order(m);
        for (long long i = 0; i < m; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](long long a, long long b) { return hi[a] < hi[b]; });
        for (long long i = 0; i < m; ++i) {
            index[i] = hi[order[i]];
            dchi_out[i] = hd[order[i]];
            for (int a = 0; a < M; ++a) amp_out[i * M + a] = ha[order[i] * M + a];
        }
    }
    if (cnt > cap) {
        ofx_set_error("ofx_trigger_find: %lld triggers found, capacity %lld", cnt, cap);
        return OFX_ERR_ARG;
    }
    return OFX_OK;
}


// ------------------------------------------------------------------------------------
// Pieces of find_triggers_once(dynamic=True) and find_triggers(residual=True)
// (oftrigger.py:78-143, 752-845, 982-986).  The dynamic pile-up window is a user-supplied
// Python function of the running range maximum, so the segmentation itself runs on the host
// over the compacted list of above-threshold samples produced here.
// ------------------------------------------------------------------------------------
namespace {

struct AboveThr {
    const float* d;
    float thr;
    __host__ __device__ bool operator()(const long long& i) const { return d[i] > thr; }
};

__global__ void k_gather(const float* __restrict__ filt, const float* __restrict__ dchi,
                         const long long* __restrict__ idx, long long m, long long n, int M,
                         float* __restrict__ oamp, float* __restrict__ odchi) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const long long j = idx[i];
    const bool in = j >= 0 && j < n;
    if (odchi) odchi[i] = in ? dchi[j] : 0.0f;
    for (int a = 0; a < M; ++a) oamp[i * M + a] = in ? filt[(size_t)a * n + j] : 0.0f;
}

// One workgroup per trigger: pulse[z] = sum_ab A_a A_b G_ab[z] with A = filtered[:, ti]
// (the delta-chi2 trace of the best-fit pulse), j = first arg-max of it, then
// dchi[ti - j + z] -= pulse[z]   (oftrigger.py:788-815).  Overlapping pulses of
// neighbouring triggers meet in atomic adds.
__global__ __launch_bounds__(TB) void k_residual(const float* __restrict__ filt,
                                                 float* __restrict__ dchi,
                                                 const float* __restrict__ G,
                                                 const long long* __restrict__ trig, long long n,
                                                 int N, int M) {
    __shared__ unsigned long long best;
    const long long ti = trig[blockIdx.x];
    if (ti < 0 || ti >= n) return;
    float A[TRIG_MAX];
    for (int a = 0; a < M; ++a) A[a] = filt[(size_t)a * n + ti];
    auto pulse = [&](int z) {
        float p = 0.0f;
        for (int a = 0; a < M; ++a)
            for (int b = 0; b < M; ++b) p = fmaf(A[a] * A[b], G[((size_t)a * M + b) * N + z], p);
        return p;
    };
    if (threadIdx.x == 0) best = 0ull;
    __syncthreads();
    unsigned long long k = 0ull;
    for (int z = threadIdx.x; z < N; z += TB) {
        const float p = pulse(z);
        // order-preserving key of a float (pulses may dip below zero between lobes)
        unsigned u = __float_as_uint(p);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        const unsigned long long c = ((unsigned long long)u << 32) |
                                     (unsigned long long)(0xFFFFFFFFu - (unsigned)z);
        k = c > k ? c : k;
    }
    atomicMax(&best, k);
    __syncthreads();
    const long long j = (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    const long long start = ti - j;
    for (int z = threadIdx.x; z < N; z += TB) {
        const long long pos = start + z;
        if (pos >= 0 && pos < n) atomicAdd(&dchi[pos], -pulse(z));
    }
}

}  // namespace

extern "C" int ofx_trigger_above(ofx_trigger* t, double chi2_threshold, long long* index,
                                 float* dchi_out, long long cap, long long* n_out, void* stream) {
    if (!t || t->n == 0) {
        ofx_set_error("ofx_trigger_above: no trace (call ofx_trigger_update_trace first)");
        return OFX_ERR_STATE;
    }
    if (!n_out || cap < 0 || (cap > 0 && (!index || !dchi_out))) {
        ofx_set_error("ofx_trigger_above: bad argument");
        return OFX_ERR_ARG;
    }
    if (t->n > 0x7fffffffLL) {       // hipcub::DeviceSelect takes an int count
        ofx_set_error("ofx_trigger_above: stream of %lld samples (the dynamic-window search handles up "
                      "to 2^31 - 1 per call: split the stream)", (long long)t->n);
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    const long long n = t->n;
    int rc;
    if ((rc = grow(&t->d_sel, &t->sel_elems, (size_t)n))) return rc;
    AboveThr pred{t->d_dchi, (float)chi2_threshold};
    hipcub::CountingInputIterator<long long> it(0);
    size_t need = 0;
    OFX_HIP(hipcub::DeviceSelect::If(nullptr, need, it, t->d_sel, t->d_count, (int)n, pred, st));
    if (need > t->tmp_bytes) {
        if (t->d_tmp) (void)hipFree(t->d_tmp);
        t->d_tmp = nullptr;
        t->tmp_bytes = 0;
        OFX_HIP(hipMalloc(&t->d_tmp, need));
        t->tmp_bytes = need;
    }
    size_t tb = t->tmp_bytes;
    OFX_HIP(hipcub::DeviceSelect::If(t->d_tmp, tb, it, t->d_sel, t->d_count, (int)n, pred, st));
    long long cnt = 0;
    OFX_HIP(hipMemcpyAsync(&cnt, t->d_count, sizeof(long long), hipMemcpyDeviceToHost, st));
    OFX_HIP(hipStreamSynchronize(st));
    *n_out = cnt;
    const long long m = std::min(cnt, cap);
    if (m > 0) {
        if ((rc = grow(&t->d_selv, &t->selv_elems, (size_t)m))) return rc;
        hipLaunchKernelGGL(k_gather, dim3(blocks_for(m)), dim3(TB), 0, st, t->d_filt, t->d_dchi,
                           t->d_sel, m, n, 0, (float*)nullptr, t->d_selv);
        OFX_HIP(hipMemcpyAsync(index, t->d_sel, (size_t)m * sizeof(long long),
                               hipMemcpyDeviceToHost, st));
        OFX_HIP(hipMemcpyAsync(dchi_out, t->d_selv, (size_t)m * sizeof(float),
                               hipMemcpyDeviceToHost, st));
        OFX_HIP(hipStreamSynchronize(st));
    }
    if (cnt > cap) {
        ofx_set_error("ofx_trigger_above: %lld samples above threshold, capacity %lld", cnt, cap);
        return OFX_ERR_ARG;
    }
    return OFX_OK;
}

extern "C" int ofx_trigger_gather(ofx_trigger* t, const long long* index, long long m,
                                  float* amplitude, float* delta_chi2, void* stream) {
    if (!t || t->n == 0 || m < 0 || (m > 0 && (!index || !amplitude))) {
        ofx_set_error("ofx_trigger_gather: bad argument / no trace");
        return OFX_ERR_ARG;
    }
    if (m == 0) return OFX_OK;
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = grow(&t->d_sel, &t->sel_elems, (size_t)m))) return rc;
    if ((rc = grow(&t->d_selv, &t->selv_elems, (size_t)m * (t->M + 1)))) return rc;
    OFX_HIP(hipMemcpyAsync(t->d_sel, index, (size_t)m * sizeof(long long), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather, dim3(blocks_for(m)), dim3(TB), 0, st, t->d_filt, t->d_dchi,
                       t->d_sel, m, t->n, t->M, t->d_selv, t->d_selv + (size_t)m * t->M);
    OFX_HIP(hipMemcpyAsync(amplitude, t->d_selv, (size_t)m * t->M * sizeof(float),
                           hipMemcpyDeviceToHost, st));
    if (delta_chi2)
        OFX_HIP(hipMemcpyAsync(delta_chi2, t->d_selv + (size_t)m * t->M, (size_t)m * sizeof(float),
                               hipMemcpyDeviceToHost, st));
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

extern "C" int ofx_trigger_set_pulse_table(ofx_trigger* t, const double* G) {
    if (!t || !G) {
        ofx_set_error("ofx_trigger_set_pulse_table: bad argument");
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(t->device));
    const size_t cnt = (size_t)t->M * t->M * t->N;
    std::vector<float> g(cnt);
    for (size_t i = 0; i < cnt; ++i) g[i] = (float)G[i];
    if (!t->d_pulse) OFX_HIP(hipMalloc(&t->d_pulse, cnt * sizeof(float)));
    OFX_HIP(hipMemcpy(t->d_pulse, g.data(), cnt * sizeof(float), hipMemcpyHostToDevice));
    return OFX_OK;
}

extern "C" int ofx_trigger_residual_subtract(ofx_trigger* t, const long long* trigger_index,
                                             long long m, void* stream) {
    if (!t || t->n == 0 || m < 0 || (m > 0 && !trigger_index)) {
        ofx_set_error("ofx_trigger_residual_subtract: bad argument / no trace");
        return OFX_ERR_ARG;
    }
    if (!t->d_pulse) {
        ofx_set_error("ofx_trigger_residual_subtract: no pulse table (ofx_trigger_set_pulse_table)");
        return OFX_ERR_STATE;
    }
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!t->saved) {                    // keep the first-pass trace (oftrigger.py:771, 826-828)
        if ((rc = grow(&t->d_dchi_saved, &t->saved_elems, (size_t)t->n))) return rc;
        OFX_HIP(hipMemcpyAsync(t->d_dchi_saved, t->d_dchi, (size_t)t->n * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
        t->saved = true;
    }
    if (m == 0) return OFX_OK;
    if ((rc = grow(&t->d_sel, &t->sel_elems, (size_t)m))) return rc;
    OFX_HIP(hipMemcpyAsync(t->d_sel, trigger_index, (size_t)m * sizeof(long long),
                           hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_residual, dim3((unsigned)m), dim3(TB), 0, st, t->d_filt, t->d_dchi,
                       t->d_pulse, t->d_sel, t->n, t->N, t->M);
    OFX_HIP(hipGetLastError());
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

extern "C" int ofx_trigger_residual_restore(ofx_trigger* t, float* residual_delta_chi2, int mem,
                                            void* stream) {
    if (!t || t->n == 0) {
        ofx_set_error("ofx_trigger_residual_restore: no trace");
        return OFX_ERR_STATE;
    }
    if (!t->saved) {
        // nothing to put back (the subtract call failed before it saved the trace): the first-pass
        // trace is still in place; a caller that asked for the residual trace gets an error
        if (residual_delta_chi2) {
            ofx_set_error("ofx_trigger_residual_restore: nothing saved");
            return OFX_ERR_STATE;
        }
        return OFX_OK;
    }
    OFX_HIP(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    if (residual_delta_chi2)
        OFX_HIP(hipMemcpyAsync(residual_delta_chi2, t->d_dchi, (size_t)t->n * sizeof(float),
                               mem == OFX_MEM_HOST ? hipMemcpyDeviceToHost
                                                   : hipMemcpyDeviceToDevice, st));
    OFX_HIP(hipMemcpyAsync(t->d_dchi, t->d_dchi_saved, (size_t)t->n * sizeof(float),
                           hipMemcpyDeviceToDevice, st));
    OFX_HIP(hipStreamSynchronize(st));
    t->saved = false;
    return OFX_OK;
}
