// ofx_nxm.hip -- N-channel x M-template optimal filter (include/ofx.h, "ofx_nxm";
// FeatureExtractors.ofnxm, detprocess/core/algorithms.py:141-274).
//
// Per event: N real transforms done as complex transforms of half the length on the packed
// traces (z[m] = x[2m] + i x[2m+1], rocFFT C2C), one pass that unpacks the real spectra of the
// pair (k, N/2 - k), folds the N spectra into the M filtered spectra Q_m = sum_b phi_mb V_b,
// accumulates chi2_0 and repacks for the inverse (k_nxm_mid), M inverse transforms, and one
// reduction per event that scans the rolled window for the maximum
// of q^T P^-1 q and writes amplitudes, t0 and chi2 (k_nxm_search).  All four passes are
// HBM-bound streaming work; the small per-bin matrix products (N, M <= 4) stay on the VALU.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "ofx_device.h"

#define NXM_MAX 4
#define NXM_MAX_SEARCHES 8

struct NxmSearch {
    int kind;          // OFX_SEARCH_NODELAY / OFX_SEARCH_DELAY
    int lo, hi;        // half-open rolled range
    int outside;
    int interp;        // OFX_SEARCH_DELAY_INTERP: interpolate_t0 (algorithms.py:152, 259)
};

struct NxmParams {
    int N, K, pre, nblk, row, n_search;
    float inv_fs;
    float pinv[NXM_MAX * NXM_MAX];
    NxmSearch search[NXM_MAX_SEARCHES];
};

struct NxmFft {
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
};

struct ofx_nxm {
    int N = 0, K = 0, pre = 0, C = 0, M = 0, max_batch = 0, device = 0;
    double fs = 0;
    int n_total = 0;
    int idx[NXM_MAX] = {0, 1, 2, 3};
    bool filter_set = false;
    float2* d_phi = nullptr;    // [M][C][K]  phi / (N fs)
    float2* d_icov = nullptr;   // [C][C][K]  Ci w_k / (N fs), w_k = 2 (1 at DC and Nyquist)
    double pinv[NXM_MAX * NXM_MAX] = {0};
    std::vector<NxmSearch> searches;
    float* d_x = nullptr;       // [max_batch, C, N] gathered channels (when needed)
    float2* d_spec = nullptr;   // [max_batch, C, N/2] packed spectra
    float2* d_q = nullptr;      // [max_batch, M, N/2] repacked filtered spectra
    float* d_qt = nullptr;      // [max_batch, M, N]
    float* d_chi0p = nullptr;   // [max_batch, nblk]
    std::map<long long, NxmFft> fft;   // keyed by events per call
    void* d_work = nullptr;
    size_t work_bytes = 0;
    float* d_stage_in = nullptr;       // [max_batch, n_total, N]; regrown when n_total grows
    size_t stage_in_floats = 0;
    uint8_t* d_stage_valid = nullptr;
    float* d_stage_out = nullptr;
    size_t stage_out_floats = 0;
    OfxLdsFft* ldsfft = nullptr;       // non-power-of-two lengths the LDS transform handles
    bool ldsfft_tried = false;
    int16_t* d_adc = nullptr;          // ofx_nxm_process_adc: staged streams, trigger indices
    size_t adc_elems = 0;
    long long* d_trig = nullptr;
    size_t trig_elems = 0;
};

namespace {

constexpr int TB = 256;

__global__ void k_nxm_gather(const float* __restrict__ traces, int n_total, int C, int4 idx,
                             int N, float* __restrict__ x) {
    const long long e = blockIdx.y;
    const int c = blockIdx.z;
    const int ch = c == 0 ? idx.x : c == 1 ? idx.y : c == 2 ? idx.z : idx.w;
    const float2* src = reinterpret_cast<const float2*>(traces + ((size_t)e * n_total + ch) * N);
    float2* dst = reinterpret_cast<float2*>(x + ((size_t)e * C + c) * N);
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i < N / 2) dst[i] = src[i];      // N is even
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One block per event on the packed spectra Z_b = FFT_{N/2}(z_b).  For the pair (k, p = N/2 - k):
//   u = Z_k + conj(Z_p), w = Z_k - conj(Z_p), s = i t_k w, t_k = exp(-2 pi i k / N)
//   V_k = (u - s)/2, conj(V_p) = (u + s)/2                  spectrum of the real trace
//   Q_m = sum_b phi_mb V_b at k and at p;  chi2_0 += Re(V^H Ci V) at k and at p
//   Ye = Q_k + conj(Q_p), Yo = (Q_k - conj(Q_p)) conj(t_k)
//   Z'_k = Ye + i Yo, Z'_p = conj(Ye - i Yo)                inverse FFT_{N/2}: q(2n) + i q(2n+1)
// k = 0 pairs DC with Nyquist; k = N/4 (N/2 even) pairs with itself.
// A workgroup takes NXM_EV consecutive events so that the filter tables of a bin pair are
// fetched once (registers / L1) for several events.
#define NXM_EV 4
template <int C, int M>
__global__ void __launch_bounds__(TB)
k_nxm_mid(int Mh, int N, int K, long long n_events, const float2* __restrict__ phi,
          const float2* __restrict__ icov, const float2* __restrict__ spec, float2* __restrict__ q,
          float* __restrict__ chi0p) {
    __shared__ float scratch[TB / OFX_WAVE];
    const long long e0 = (long long)blockIdx.x * NXM_EV;
    float part[NXM_EV];
#pragma unroll
    for (int i = 0; i < NXM_EV; ++i) part[i] = 0.0f;
    // the workgroups start at different places of the spectrum: rows lie 4 N bytes apart, and a
    // lock-step sweep would keep every workgroup on the same few memory channels
    const int nchunk = (Mh / 2 + TB) / TB;
    const int first = (int)((blockIdx.x * 7u) % (unsigned)nchunk);
    for (int j = 0; j < nchunk; ++j) {
        int c = first + j;
        if (c >= nchunk) c -= nchunk;
        const int k = c * TB + threadIdx.x;
        if (k > Mh / 2) continue;
        const int p = (k == 0) ? 0 : Mh - k;
        const int kp = (k == 0) ? Mh : p;               // one-sided bin of the partner
        float sn, cs;
        sincospif(-2.0f * (float)k / (float)N, &sn, &cs);
        // every load of the bin pair is issued before the first use (the last workgroup
        // re-reads its last event instead of branching)
        float2 zks[NXM_EV][C], zps[NXM_EV][C];
#pragma unroll
        for (int i = 0; i < NXM_EV; ++i) {
            const size_t el = (size_t)(e0 + i < n_events ? e0 + i : n_events - 1);
#pragma unroll
            for (int b = 0; b < C; ++b) {
                zks[i][b] = spec[(el * C + b) * Mh + k];
                zps[i][b] = spec[(el * C + b) * Mh + p];
            }
        }
#pragma unroll
        for (int i = 0; i < NXM_EV; ++i) {
            const size_t e = (size_t)(e0 + i);
            const bool live = e0 + i < n_events;
            float2 vk[C], vp[C];                        // V_k, V_p
#pragma unroll
            for (int b = 0; b < C; ++b) {
                const float2 zk = zks[i][b], zp = zps[i][b];
                if (k == 0) {
                    vk[b] = make_float2(zk.x + zk.y, 0.0f);
                    vp[b] = make_float2(zk.x - zk.y, 0.0f);
                } else {
                    const float2 u = make_float2(zk.x + zp.x, zk.y - zp.y);
                    const float2 w = make_float2(zk.x - zp.x, zk.y + zp.y);
                    const float2 sv = make_float2(-(cs * w.y + sn * w.x), cs * w.x - sn * w.y);
                    vk[b] = make_float2(0.5f * (u.x - sv.x), 0.5f * (u.y - sv.y));
                    vp[b] = make_float2(0.5f * (u.x + sv.x), -0.5f * (u.y + sv.y));
                }
            }
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < C; ++a) {
                float2 rk = make_float2(0.0f, 0.0f), rp = make_float2(0.0f, 0.0f);
#pragma unroll
                for (int b = 0; b < C; ++b) {
                    const float2 tk = cmulf(icov[((size_t)a * C + b) * K + k], vk[b]);
                    const float2 tp = cmulf(icov[((size_t)a * C + b) * K + kp], vp[b]);
                    rk.x += tk.x; rk.y += tk.y;
                    rp.x += tp.x; rp.y += tp.y;
                }
                acc += vk[a].x * rk.x + vk[a].y * rk.y;
                if (kp != k) acc += vp[a].x * rp.x + vp[a].y * rp.y;
            }
            part[i] += acc;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float2 qk = make_float2(0.0f, 0.0f), qp = make_float2(0.0f, 0.0f);
#pragma unroll
                for (int b = 0; b < C; ++b) {
                    const float2 tk = cmulf(phi[((size_t)m * C + b) * K + k], vk[b]);
                    const float2 tp = cmulf(phi[((size_t)m * C + b) * K + kp], vp[b]);
                    qk.x += tk.x; qk.y += tk.y;
                    qp.x += tp.x; qp.y += tp.y;
                }
                if (k == 0) qk.y = qp.y = 0.0f;   // DC and Nyquist of a real sequence: real parts only
                const float2 ye = make_float2(qk.x + qp.x, qk.y - qp.y);      // Q_k + conj(Q_p)
                const float2 d = make_float2(qk.x - qp.x, qk.y + qp.y);       // Q_k - conj(Q_p)
                const float2 yo = make_float2(d.x * cs + d.y * sn, d.y * cs - d.x * sn);
                float2* zo = q + (e * M + m) * Mh;
                if (live) zo[k] = make_float2(ye.x - yo.y, ye.y + yo.x);
                if (live && p != k) zo[p] = make_float2(ye.x + yo.y, -(ye.y - yo.x));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NXM_EV; ++i) {
        const float sum = ofx_block_sum(part[i], scratch);
        if (threadIdx.x == 0 && e0 + i < n_events) chi0p[e0 + i] = sum;
    }
}

template <int M>
__device__ __forceinline__ float nxm_quad(const float (&qv)[M], const float* pinv) {
    float r = 0.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float t = 0.0f;
#pragma unroll
        for (int l = 0; l < M; ++l) t += pinv[m * M + l] * qv[l];
        r += qv[m] * t;
    }
    return r;
}

// one workgroup per event: chi2_0, then per search the arg-max of q^T P^-1 q over the rolled
// window (ties -> smallest rolled index = NumPy argmin of chi2 on the rolled array)
template <int M>
__global__ void __launch_bounds__(TB)
k_nxm_search(NxmParams p, const float* __restrict__ qt, const float* __restrict__ chi0p,
             const uint8_t* __restrict__ valid, float* __restrict__ out) {
    __shared__ float fscratch[TB / OFX_WAVE];
    __shared__ OfxCand cscratch[TB / OFX_WAVE];
    const long long e = blockIdx.x;
    float* row = out + (size_t)e * p.row;
    if (valid && !valid[e]) {
        for (int i = threadIdx.x; i < p.row; i += TB) row[i] = OFX_SENTINEL;
        return;
    }
    const float* qe = qt + (size_t)e * M * p.N;
    float c = 0.0f;
    for (int j = threadIdx.x; j < p.nblk; j += TB) c += chi0p[(size_t)e * p.nblk + j];
    const float chi0 = ofx_block_sum(c, fscratch);
    for (int s = 0; s < p.n_search; ++s) {
        const NxmSearch sr = p.search[s];
        float* rec = row + s * (M + 3);
        int best = -1;
        if (sr.kind == OFX_SEARCH_NODELAY) {
            best = p.pre;
        } else {
            OfxCand cand;
            cand.key = -INFINITY;
            cand.idx = 0x7fffffff;
            cand.amp = 0.0f;
            const int n_in = sr.hi - sr.lo;
            const int total = sr.outside ? p.N - n_in : n_in;
            for (int j = threadIdx.x; j < total; j += TB) {
                const int i = sr.outside ? (j < sr.lo ? j : j + n_in) : sr.lo + j;
                int n = i - p.pre;
                if (n < 0) n += p.N;
                float qv[M];
#pragma unroll
                for (int m = 0; m < M; ++m) qv[m] = qe[(size_t)m * p.N + n];
                const float r = nxm_quad<M>(qv, p.pinv);
                if (ofx_cand_better(r, i, cand)) {
                    cand.key = r;
                    cand.idx = i;
                }
            }
            cand = ofx_cand_block_reduce(cand, cscratch);
            if (total > 0) best = cand.idx;
        }
        if (best < 0) {
            if (threadIdx.x < M + 3) rec[threadIdx.x] = OFX_SENTINEL;
            continue;
        }
        if (threadIdx.x == 0) {
            int n = best - p.pre;
            if (n < 0) n += p.N;
            float qv[M];
#pragma unroll
            for (int m = 0; m < M; ++m) qv[m] = qe[(size_t)m * p.N + n];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float t = 0.0f;
#pragma unroll
                for (int l = 0; l < M; ++l) t += p.pinv[m * M + l] * qv[l];
                rec[m] = t;
            }
            const float r0 = nxm_quad<M>(qv, p.pinv);
            float frac = 0.0f, chi2 = chi0 - r0;
            if (sr.interp && best > 0 && best < p.N - 1) {
                // interpolate_t0: vertex of the parabola through chi2 = chi0 - q^T P^-1 q at
                // the rolled bins best-1, best, best+1; every amplitude from its own parabola
                // at the same offset (the of1x1 rule, ofx_interpolate / oracle interpolate_of)
                float qm[M], qp[M], am[M], ap[M];
                const int nm = (n == 0) ? p.N - 1 : n - 1, np_ = (n == p.N - 1) ? 0 : n + 1;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    qm[m] = qe[(size_t)m * p.N + nm];
                    qp[m] = qe[(size_t)m * p.N + np_];
                }
                const float rm = nxm_quad<M>(qm, p.pinv), rp = nxm_quad<M>(qp, p.pinv);
                const float dpm = rp - rm;                       // y(-1) - y(+1)
                const float den = (r0 - rm) + (r0 - rp);         // y(-1) - 2 y(0) + y(+1)
                if (den > 0.0f) {
                    const float x = 0.5f * dpm / den;
                    if (fabsf(x) <= 1.0f) {
                        frac = x;
                        chi2 = chi2 - 0.125f * dpm * dpm / den;
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            float tm = 0.0f, tp = 0.0f;
#pragma unroll
                            for (int l = 0; l < M; ++l) {
                                tm += p.pinv[m * M + l] * qm[l];
                                tp += p.pinv[m * M + l] * qp[l];
                            }
                            am[m] = tm;
                            ap[m] = tp;
                            const float a0 = rec[m];
                            rec[m] = a0 + 0.5f * (ap[m] - am[m]) * x +
                                     0.5f * ((am[m] - a0) + (ap[m] - a0)) * x * x;
                        }
                    }
                }
            }
            rec[M] = ((float)(best - p.pre) + frac) * p.inv_fs;
            rec[M + 1] = chi2;
            rec[M + 2] = (float)best;
        }
    }
}

template <typename T>
int grow_to(T** buf, size_t elems) {
    if (*buf) return OFX_OK;
    OFX_HIP(hipMalloc(reinterpret_cast<void**>(buf), elems * sizeof(T)));
    return OFX_OK;
}

// staging buffer of the events: its size follows n_total, which ofx_nxm_set_channels may change
int ensure_stage_in(ofx_nxm* p, size_t floats, hipStream_t st) {
    if (p->stage_in_floats >= floats) return OFX_OK;
    OFX_HIP(hipStreamSynchronize(st));
    if (p->d_stage_in) (void)hipFree(p->d_stage_in);
    p->d_stage_in = nullptr;
    p->stage_in_floats = 0;
    OFX_HIP(hipMalloc(&p->d_stage_in, floats * sizeof(float)));
    p->stage_in_floats = floats;
    return OFX_OK;
}

int get_fft(ofx_nxm* p, long long nb, hipStream_t st, NxmFft** out) {
    auto it = p->fft.find(nb);
    if (it == p->fft.end()) {
        {
            const int rc_setup = ofx_rocfft_setup_once();
            if (rc_setup) return rc_setup;
        }
        NxmFft f;
        const size_t len = (size_t)p->N / 2;             // packed complex points
        OFX_FFT(rocfft_plan_create(&f.fwd, rocfft_placement_notinplace,
                                   rocfft_transform_type_complex_forward, rocfft_precision_single,
                                   1, &len, (size_t)nb * p->C, nullptr));
        OFX_FFT(rocfft_plan_create(&f.inv, rocfft_placement_notinplace,
                                   rocfft_transform_type_complex_inverse, rocfft_precision_single,
                                   1, &len, (size_t)nb * p->M, nullptr));
        size_t w1 = 0, w2 = 0;
        OFX_FFT(rocfft_plan_get_work_buffer_size(f.fwd, &w1));
        OFX_FFT(rocfft_plan_get_work_buffer_size(f.inv, &w2));
        const size_t wb = w1 > w2 ? w1 : w2;
        if (wb > p->work_bytes) {
            OFX_HIP(hipStreamSynchronize(st));
            if (p->d_work) (void)hipFree(p->d_work);
            p->d_work = nullptr;
            p->work_bytes = 0;
            OFX_HIP(hipMalloc(&p->d_work, wb));
            p->work_bytes = wb;
        }
        OFX_FFT(rocfft_execution_info_create(&f.info));
        it = p->fft.emplace(nb, f).first;
    }
    NxmFft& f = it->second;
    if (p->work_bytes)
        OFX_FFT(rocfft_execution_info_set_work_buffer(f.info, p->d_work, p->work_bytes));
    OFX_FFT(rocfft_execution_info_set_stream(f.info, st));
    *out = &f;
    return OFX_OK;
}

template <int C>
void launch_mid(ofx_nxm* p, long long nb, hipStream_t st) {
    const dim3 grid((unsigned)((nb + NXM_EV - 1) / NXM_EV));
#define MID(MM)                                                                            \
    hipLaunchKernelGGL((k_nxm_mid<C, MM>), grid, dim3(TB), 0, st, p->N / 2, p->N, p->K, nb, \
                       p->d_phi, p->d_icov, p->d_spec, p->d_q, p->d_chi0p)
    switch (p->M) {
        case 1: MID(1); break;
        case 2: MID(2); break;
        case 3: MID(3); break;
        default: MID(4); break;
    }
#undef MID
}

int process_device(ofx_nxm* p, const float* traces, const uint8_t* valid, long long nb,
                   float* out, hipStream_t st) {
    const int N = p->N, K = p->K, C = p->C, M = p->M;
    const int nblk = 1;                                  // chi2_0: one partial per event
    const size_t mb = (size_t)p->max_batch;
    int rc;
    (void)K;
    if ((rc = grow_to(&p->d_spec, mb * C * (N / 2)))) return rc;
    if ((rc = grow_to(&p->d_q, mb * M * (N / 2)))) return rc;
    if ((rc = grow_to(&p->d_qt, mb * M * N))) return rc;
    if ((rc = grow_to(&p->d_chi0p, mb * nblk))) return rc;
    bool identity = p->n_total == C;
    for (int c = 0; c < C; ++c) identity = identity && p->idx[c] == c;
    const float* x = traces;
    if (!identity) {
        if ((rc = grow_to(&p->d_x, mb * C * N))) return rc;
        const int4 idx = make_int4(p->idx[0], p->idx[1], p->idx[2], p->idx[3]);
        hipLaunchKernelGGL(k_nxm_gather, dim3((N / 2 + TB - 1) / TB, (unsigned)nb, C), dim3(TB), 0,
                           st, traces, p->n_total, C, idx, N, p->d_x);
        x = p->d_x;
    }
    // transforms: rocFFT, except for non-power-of-two lengths of the form 2^a 3^b 5^c that fit in
    // LDS, where rocFFT takes a multi-kernel path and the one-kernel LDS transform is faster
    if (!p->ldsfft_tried) {
        p->ldsfft_tried = true;
        const int Mh = N / 2;
        if ((Mh & (Mh - 1)) != 0 || Mh == 16384) {     // 16384: the register-resident transform
            const int r = ofx_ldsfft_create(Mh, p->device, &p->ldsfft, true);
            if (r != OFX_OK && r != OFX_ERR_UNSUPPORTED) return r;
        }
    }
    NxmFft* f = nullptr;
    if (p->ldsfft) {
        if ((rc = ofx_ldsfft_exec(p->ldsfft, true, reinterpret_cast<const float2*>(x), p->d_spec,
                                  nb * C, st)))
            return rc;
    } else {
        if ((rc = get_fft(p, nb, st, &f))) return rc;
        void* in1[1] = {(void*)x};
        void* out1[1] = {(void*)p->d_spec};
        OFX_FFT(rocfft_execute(f->fwd, in1, out1, f->info));
    }
    switch (C) {
        case 1: launch_mid<1>(p, nb, st); break;
        case 2: launch_mid<2>(p, nb, st); break;
        case 3: launch_mid<3>(p, nb, st); break;
        default: launch_mid<4>(p, nb, st); break;
    }
    if (p->ldsfft) {
        if ((rc = ofx_ldsfft_exec(p->ldsfft, false, p->d_q, reinterpret_cast<float2*>(p->d_qt),
                                  nb * M, st)))
            return rc;
    } else {
        void* in2[1] = {(void*)p->d_q};
        void* out2[1] = {(void*)p->d_qt};
        OFX_FFT(rocfft_execute(f->inv, in2, out2, f->info));
    }
    NxmParams prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.N = N;
    prm.K = K;
    prm.pre = p->pre;
    prm.nblk = nblk;
    prm.n_search = (int)p->searches.size();
    prm.row = prm.n_search * (M + 3);
    prm.inv_fs = (float)(1.0 / p->fs);
    for (int i = 0; i < M * M; ++i) prm.pinv[i] = (float)p->pinv[i];
    for (int s = 0; s < prm.n_search; ++s) prm.search[s] = p->searches[s];
#define SRCH(MM)                                                                             \
    hipLaunchKernelGGL((k_nxm_search<MM>), dim3((unsigned)nb), dim3(TB), 0, st, prm, p->d_qt, \
                       p->d_chi0p, valid, out)
    switch (M) {
        case 1: SRCH(1); break;
        case 2: SRCH(2); break;
        case 3: SRCH(3); break;
        default: SRCH(4); break;
    }
#undef SRCH
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

}  // namespace

extern "C" int ofx_nxm_create(ofx_nxm** out, int n_samples, int n_pretrigger, double fs,
                              int n_chan, int n_tmpl, int max_batch, int device) {
    if (!out || n_samples < 4 || (n_samples & 1) || n_pretrigger < 0 ||
        n_pretrigger >= n_samples || !(fs > 0) || n_chan < 1 || n_chan > NXM_MAX || n_tmpl < 1 ||
        n_tmpl > NXM_MAX || max_batch < 1) {
        ofx_set_error("ofx_nxm_create: bad argument (even n_samples, 1..%d channels and templates)",
                      NXM_MAX);
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(device));
    ofx_nxm* p = new ofx_nxm();
    p->N = n_samples;
    p->K = n_samples / 2 + 1;
    p->pre = n_pretrigger;
    p->fs = fs;
    p->C = n_chan;
    p->M = n_tmpl;
    p->n_total = n_chan;
    p->max_batch = max_batch > 65535 ? 65535 : max_batch;
    p->device = device;
    *out = p;
    return OFX_OK;
}

extern "C" int ofx_nxm_destroy(ofx_nxm* p) {
    if (!p) return OFX_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : p->fft) {
        if (kv.second.fwd) rocfft_plan_destroy(kv.second.fwd);
        if (kv.second.inv) rocfft_plan_destroy(kv.second.inv);
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
    }
    ofx_ldsfft_destroy(p->ldsfft);
    void* bufs[] = {p->d_phi, p->d_icov, p->d_x, p->d_spec, p->d_q, p->d_qt, p->d_chi0p,
                    p->d_work, p->d_stage_in, p->d_stage_valid, p->d_stage_out, p->d_adc,
                    p->d_trig};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete p;
    return OFX_OK;
}

extern "C" int ofx_nxm_set_filter(ofx_nxm* p, const double* phi, const double* icov,
                                  const double* pinv) {
    if (!p || !phi || !icov || !pinv) {
        ofx_set_error("ofx_nxm_set_filter: bad argument");
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(p->device));
    const int K = p->K, C = p->C, M = p->M;
    const double sc = 1.0 / ((double)p->N * p->fs);
    std::vector<float2> hp((size_t)M * C * K), hc((size_t)C * C * K);
    for (size_t r = 0; r < (size_t)M * C; ++r)
        for (int k = 0; k < K; ++k) {
            const double* v = phi + 2 * (r * K + k);
            hp[r * K + k] = make_float2((float)(v[0] * sc), (float)(v[1] * sc));
        }
    for (size_t r = 0; r < (size_t)C * C; ++r)
        for (int k = 0; k < K; ++k) {
            const double w = (k == 0 || k == p->N / 2) ? sc : 2.0 * sc;
            const double* v = icov + 2 * (r * K + k);
            hc[r * K + k] = make_float2((float)(v[0] * w), (float)(v[1] * w));
        }
    OFX_HIP(hipDeviceSynchronize());
    if (!p->d_phi) OFX_HIP(hipMalloc(&p->d_phi, hp.size() * sizeof(float2)));
    if (!p->d_icov) OFX_HIP(hipMalloc(&p->d_icov, hc.size() * sizeof(float2)));
    OFX_HIP(hipMemcpy(p->d_phi, hp.data(), hp.size() * sizeof(float2), hipMemcpyHostToDevice));
    OFX_HIP(hipMemcpy(p->d_icov, hc.data(), hc.size() * sizeof(float2), hipMemcpyHostToDevice));
    for (int i = 0; i < M * M; ++i) p->pinv[i] = pinv[i];
    p->filter_set = true;
    return OFX_OK;
}

extern "C" int ofx_nxm_add_search(ofx_nxm* p, int kind, int lo, int hi, int outside) {
    if (!p || (kind != OFX_SEARCH_NODELAY && kind != OFX_SEARCH_DELAY &&
               kind != OFX_SEARCH_DELAY_INTERP)) {
        ofx_set_error("ofx_nxm_add_search: kind must be OFX_SEARCH_NODELAY, OFX_SEARCH_DELAY or "
                      "OFX_SEARCH_DELAY_INTERP");
        return -OFX_ERR_ARG;
    }
    if ((int)p->searches.size() >= NXM_MAX_SEARCHES) {
        ofx_set_error("ofx_nxm_add_search: at most %d searches", NXM_MAX_SEARCHES);
        return -OFX_ERR_ARG;
    }
    NxmSearch s;
    s.interp = (kind == OFX_SEARCH_DELAY_INTERP) ? 1 : 0;
    s.kind = s.interp ? OFX_SEARCH_DELAY : kind;
    s.lo = lo < 0 ? 0 : lo;
    s.hi = hi > p->N ? p->N : hi;
    if (s.hi < s.lo) s.hi = s.lo;
    s.outside = outside ? 1 : 0;
    p->searches.push_back(s);
    return (int)p->searches.size() - 1;
}

extern "C" int ofx_nxm_reset_searches(ofx_nxm* p) {
    if (!p) return OFX_ERR_ARG;
    p->searches.clear();
    return OFX_OK;
}

extern "C" int ofx_nxm_set_channels(ofx_nxm* p, int n_channels_total, const int* index) {
    if (!p || !index || n_channels_total < p->C) {
        ofx_set_error("ofx_nxm_set_channels: bad argument");
        return OFX_ERR_ARG;
    }
    for (int c = 0; c < p->C; ++c)
        if (index[c] < 0 || index[c] >= n_channels_total) {
            ofx_set_error("ofx_nxm_set_channels: channel index %d out of range", index[c]);
            return OFX_ERR_ARG;
        }
    for (int c = 0; c < NXM_MAX; ++c) p->idx[c] = c < p->C ? index[c] : 0;
    p->n_total = n_channels_total;
    return OFX_OK;
}

extern "C" int ofx_nxm_row_floats(const ofx_nxm* p) {
    return p ? (int)p->searches.size() * (p->M + 3) : 0;
}

extern "C" int ofx_nxm_process(ofx_nxm* p, const float* traces, const uint8_t* valid,
                               long long n, int traces_mem, float* out, int out_mem,
                               void* stream) {
    if (!p || n < 0 || (n > 0 && (!traces || !out))) {
        ofx_set_error("ofx_nxm_process: bad argument");
        return OFX_ERR_ARG;
    }
    if (!p->filter_set || p->searches.empty()) {
        ofx_set_error("ofx_nxm_process: no filter or no search set");
        return OFX_ERR_STATE;
    }
    if (n == 0) return OFX_OK;
    OFX_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t ev_floats = (size_t)p->n_total * p->N;
    const int row = ofx_nxm_row_floats(p);
    const long long chunk = p->max_batch;
    int rc;
    for (long long b0 = 0; b0 < n; b0 += chunk) {
        const long long nb = (n - b0 < chunk) ? (n - b0) : chunk;
        const float* d_in = traces + (size_t)b0 * ev_floats;
        const uint8_t* d_valid = valid ? valid + b0 : nullptr;
        float* d_out = out + (size_t)b0 * row;
        if (traces_mem == OFX_MEM_HOST) {
            if ((rc = ensure_stage_in(p, (size_t)chunk * ev_floats, st))) return rc;
            OFX_HIP(hipMemcpyAsync(p->d_stage_in, d_in, (size_t)nb * ev_floats * sizeof(float),
                                   hipMemcpyHostToDevice, st));
            d_in = p->d_stage_in;
            if (valid) {
                if ((rc = grow_to(&p->d_stage_valid, (size_t)chunk))) return rc;
                OFX_HIP(hipMemcpyAsync(p->d_stage_valid, valid + b0, (size_t)nb,
                                       hipMemcpyHostToDevice, st));
                d_valid = p->d_stage_valid;
            }
        }
        if (out_mem == OFX_MEM_HOST) {
            const size_t want = (size_t)chunk * row;
            if (p->stage_out_floats < want) {
                OFX_HIP(hipStreamSynchronize(st));
                if (p->d_stage_out) (void)hipFree(p->d_stage_out);
                p->d_stage_out = nullptr;
                p->stage_out_floats = 0;
                OFX_HIP(hipMalloc(&p->d_stage_out, want * sizeof(float)));
                p->stage_out_floats = want;
            }
            d_out = p->d_stage_out;
        }
        if ((rc = process_device(p, d_in, d_valid, nb, d_out, st))) return rc;
        if (out_mem == OFX_MEM_HOST)
            OFX_HIP(hipMemcpyAsync(out + (size_t)b0 * row, d_out, (size_t)nb * row * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        if (traces_mem == OFX_MEM_HOST || out_mem == OFX_MEM_HOST)
            OFX_HIP(hipStreamSynchronize(st));
    }
    return OFX_OK;
}

extern "C" int ofx_nxm_process_adc(ofx_nxm* p, const int16_t* adc, long long n_stream, int adc_mem,
                                   const long long* trigger_index, long long n,
                                   const double* scale, const double* offset, float* out,
                                   int out_mem, void* stream) {
    if (!p || n < 0 || n_stream < 0 || (n > 0 && (!adc || !trigger_index || !out)) || !scale ||
        !offset) {
        ofx_set_error("ofx_nxm_process_adc: bad argument");
        return OFX_ERR_ARG;
    }
    if (!p->filter_set || p->searches.empty()) {
        ofx_set_error("ofx_nxm_process_adc: no filter or no search set");
        return OFX_ERR_STATE;
    }
    if (n == 0) return OFX_OK;
    OFX_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const int CT = p->n_total;
    const size_t ev_floats = (size_t)CT * p->N;
    const int row = ofx_nxm_row_floats(p);
    const int16_t* d_adc = adc;
    if (adc_mem == OFX_MEM_HOST) {
        const size_t want = (size_t)CT * (size_t)n_stream;
        if (p->adc_elems < want) {
            OFX_HIP(hipStreamSynchronize(st));
            if (p->d_adc) (void)hipFree(p->d_adc);
            p->d_adc = nullptr;
            p->adc_elems = 0;
            OFX_HIP(hipMalloc(&p->d_adc, want * sizeof(int16_t)));
            p->adc_elems = want;
        }
        OFX_HIP(hipMemcpyAsync(p->d_adc, adc, want * sizeof(int16_t), hipMemcpyHostToDevice, st));
        d_adc = p->d_adc;
    }
    if (p->trig_elems < (size_t)n) {
        OFX_HIP(hipStreamSynchronize(st));
        if (p->d_trig) (void)hipFree(p->d_trig);
        p->d_trig = nullptr;
        p->trig_elems = 0;
        OFX_HIP(hipMalloc(&p->d_trig, (size_t)n * sizeof(long long)));
        p->trig_elems = (size_t)n;
    }
    OFX_HIP(hipMemcpyAsync(p->d_trig, trigger_index, (size_t)n * sizeof(long long),
                           hipMemcpyHostToDevice, st));
    std::vector<float> sc(CT), of(CT);
    for (int c = 0; c < CT; ++c) {
        sc[c] = (float)scale[c];
        of[c] = (float)offset[c];
    }
    const long long chunk = p->max_batch;
    int rc;
    if ((rc = ensure_stage_in(p, (size_t)chunk * ev_floats, st))) return rc;
    if ((rc = grow_to(&p->d_stage_valid, (size_t)chunk))) return rc;
    if (out_mem == OFX_MEM_HOST && p->stage_out_floats < (size_t)chunk * row) {
        OFX_HIP(hipStreamSynchronize(st));
        if (p->d_stage_out) (void)hipFree(p->d_stage_out);
        p->d_stage_out = nullptr;
        p->stage_out_floats = 0;
        OFX_HIP(hipMalloc(&p->d_stage_out, (size_t)chunk * row * sizeof(float)));
        p->stage_out_floats = (size_t)chunk * row;
    }
    for (long long b0 = 0; b0 < n; b0 += chunk) {
        const long long nb = (n - b0 < chunk) ? (n - b0) : chunk;
        rc = ofx_cut_launch(d_adc, n_stream, CT, p->N, p->pre, p->d_trig + b0, nb, sc.data(),
                            of.data(), p->d_stage_in, p->d_stage_valid, st);
        if (rc) return rc;
        float* d_out = (out_mem == OFX_MEM_HOST) ? p->d_stage_out : out + (size_t)b0 * row;
        if ((rc = process_device(p, p->d_stage_in, p->d_stage_valid, nb, d_out, st))) return rc;
        if (out_mem == OFX_MEM_HOST)
            OFX_HIP(hipMemcpyAsync(out + (size_t)b0 * row, d_out, (size_t)nb * row * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        OFX_HIP(hipStreamSynchronize(st));      // the staging buffers are reused by the next chunk
    }
    return OFX_OK;
}
