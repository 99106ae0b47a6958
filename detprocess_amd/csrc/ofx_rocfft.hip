// ofx_rocfft.hip -- ROCFFT engine: the unfused pipeline named by north_star
//   [prep/time-domain kernel] -> rocFFT forward -> filter-apply + chi2_0 kernel
//   -> rocFFT inverse -> arg-max / chi2 / lowchi2 kernel.
// The real trace of N samples is transformed as M = N/2 packed complex points
// (z[m] = x[2m] + i x[2m+1]) with rocFFT's complex plans, and ONE kernel (k_mid) does the
// real-FFT unpack, the filter multiply, chi2_0 and the re-pack for the inverse: rocFFT's own
// real-transform plans spend two extra full passes over the data (r2c post-, c2r
// pre-processing kernels) on exactly that algebra.
// Handles any even trace length; the FUSED engine (ofx_fused.hip) replaces it
// for the power-of-two lengths it supports.  Also the on-device cross-check of
// the fused kernel in the GPU tests.
#include <algorithm>

#include "ofx_common.h"
#include "ofx_device.h"

#define RB 256   // threads per block in this file

// ---------------------------------------------------------------------------
// prep: channel algebra on load (processing_data.py:1033-1047) + time-domain
// window features (algorithms.py:698, 759, 818, 879).  One block per event.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void k_prep(OfxPlanDev pd, const float* __restrict__ ev,
                                             const uint8_t* __restrict__ valid,
                                             float* __restrict__ combined,
                                             float* __restrict__ out) {
    __shared__ float scratch[RB / OFX_WAVE];
    const long long b = blockIdx.x;
    float* row = out + b * pd.row;
    if (valid && !valid[b]) {
        for (int w = threadIdx.x; w < pd.n_tdwin * OFX_TDWIN_FLOATS; w += RB)
            row[pd.tdw[0].out_off + w] = OFX_SENTINEL;
        return;
    }
    const float* e = ev + (size_t)b * pd.n_channels * pd.N;
    float* c = combined ? combined + (size_t)b * pd.N : nullptr;

    auto sample = [&](int n) -> float {
        float v = pd.weight[0] * e[(size_t)pd.chan[0] * pd.N + n];
        for (int j = 1; j < pd.n_terms; ++j)
            v = fmaf(pd.weight[j], e[(size_t)pd.chan[j] * pd.N + n], v);
        return v;
    };
    if (c)
        for (int n = threadIdx.x; n < pd.N; n += RB) c[n] = sample(n);

    for (int w = 0; w < pd.n_tdwin; ++w) {
        const int lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
        float s = 0.0f, sq = 0.0f, mx = -INFINITY, mn = INFINITY;
        for (int n = lo + threadIdx.x; n < hi; n += RB) {
            const float v = sample(n);
            s += v;
            sq = fmaf(v, v, sq);
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
        }
        s = ofx_block_sum(s, scratch);
        sq = ofx_block_sum(sq, scratch);
        mx = ofx_block_max(mx, scratch);
        mn = ofx_block_min(mn, scratch);
        if (threadIdx.x == 0) {
            const float first = sample(lo), last = sample(hi - 1);
            float* o = row + pd.tdw[w].out_off;
            o[OFX_TD_BASELINE] = s / (float)(hi - lo);
            o[OFX_TD_INTEGRAL] = (s - 0.5f * (first + last)) * pd.inv_fs;
            o[OFX_TD_MAXIMUM] = mx;
            o[OFX_TD_MINIMUM] = mn;
            o[OFX_TD_SUM] = s;
            o[OFX_TD_SUMSQ] = sq;
            o[OFX_TD_FIRST] = first;
            o[OFX_TD_LAST] = last;
        }
    }
}

// ---------------------------------------------------------------------------
// psd_amp bands: mean over bins [k_lo, k_hi) of sqrt(w_k |V_k|^2 / (N fs))
// (algorithms.py:1013-1038).  One block per trace.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void k_bands(OfxPlanDev pd, const float2* __restrict__ vlow,
                                              int cap, const uint8_t* __restrict__ valid,
                                              float* __restrict__ out) {
    __shared__ float scratch[RB / OFX_WAVE];
    const size_t b = blockIdx.x;
    float* row = out + b * pd.row;
    if (valid && !valid[b]) {
        for (int i = threadIdx.x; i < pd.n_bands; i += RB) row[pd.band[i].out_off] = OFX_SENTINEL;
        return;
    }
    const float2* V = vlow + b * cap;
    const float c = 1.0f / ((float)pd.N * pd.fs);
    for (int i = 0; i < pd.n_bands; ++i) {
        const int lo = pd.band[i].k_lo, hi = pd.band[i].k_hi;
        float acc = 0.0f;
        for (int k = lo + threadIdx.x; k < hi; k += RB) {
            const float2 v = V[k];
            const float w = (2 * k == pd.N) ? 1.0f : 2.0f;
            acc += sqrtf(w * c * (v.x * v.x + v.y * v.y));
        }
        acc = ofx_block_sum(acc, scratch);
        if (threadIdx.x == 0) row[pd.band[i].out_off] = acc / (float)(hi - lo);
    }
}

// ---------------------------------------------------------------------------
// middle step on the packed spectrum Z = FFT_M(z), M = N/2 (OFBase.calc_signal_filt +
// chi2_0 -- processing_data.py:771).  For the pair (k, p = M - k):
//   u = Z_k + conj(Z_p), w = Z_k - conj(Z_p), s = i t_k w, t_k = exp(-2 pi i k / N)
//   V_k = (u - s)/2, conj(V_p) = (u + s)/2             spectrum of the real trace
//   chi2_0 += w_k g_k |V_k|^2 + w_p g_p |V_p|^2
//   Y = wf V;  Ye = Y_k + conj(Y_p), Yo = (Y_k - conj(Y_p)) conj(t_k)
//   Z'_k = Ye + i Yo,  Z'_p = conj(Ye - i Yo)          inverse FFT_M gives A(2m) + i A(2m+1)
// k = 0 pairs DC with Nyquist; k = M/2 (M even) pairs with itself.  V_k for k < cap is
// kept for the low-frequency chi2 and the psd_amp bands.  One block per trace.
// wf == nullptr: spectrum extraction only (bands without a filter slot).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void k_mid(int M, int N, const float2* __restrict__ wf,
                                            const float* __restrict__ g,
                                            const float2* __restrict__ zspec,
                                            float2* __restrict__ zfilt, float* __restrict__ chi0,
                                            float2* __restrict__ vlow, int cap) {
    __shared__ float scratch[RB / OFX_WAVE];
    const size_t b = blockIdx.x;
    const float2* Z = zspec + b * M;
    float2* Zo = zfilt ? zfilt + b * M : nullptr;
    float2* VL = vlow ? vlow + b * cap : nullptr;
    float acc = 0.0f;
    for (int k = threadIdx.x; k <= M / 2; k += RB) {
        const int p = (k == 0) ? 0 : M - k;
        const float2 zk = Z[k], zp = Z[p];
        float2 vk, vpc;                                  // V_k, conj(V_p)
        float sn, cs;
        sincospif(-2.0f * (float)k / (float)N, &sn, &cs);          // t_k
        if (k == 0) {
            vk = make_float2(zk.x + zk.y, 0.0f);                    // V_0   (DC)
            vpc = make_float2(zk.x - zk.y, 0.0f);                   // V_M   (Nyquist)
        } else {
            const float2 u = make_float2(zk.x + zp.x, zk.y - zp.y);
            const float2 w = make_float2(zk.x - zp.x, zk.y + zp.y);
            // s = i t w
            const float2 sv = make_float2(-(cs * w.y + sn * w.x), cs * w.x - sn * w.y);
            vk = make_float2(0.5f * (u.x - sv.x), 0.5f * (u.y - sv.y));
            vpc = make_float2(0.5f * (u.x + sv.x), 0.5f * (u.y + sv.y));
        }
        const int kp = (k == 0) ? M : p;                 // one-sided bin of the partner
        if (VL) {
            if (k < cap) VL[k] = vk;
            if (kp < cap && kp != k) VL[kp] = make_float2(vpc.x, -vpc.y);
        }
        if (!wf) continue;
        const float wk = (k == 0) ? 1.0f : 2.0f;
        const float wp = (kp == M) ? 1.0f : 2.0f;
        acc = fmaf(wk * g[k], vk.x * vk.x + vk.y * vk.y, acc);
        if (kp != k) acc = fmaf(wp * g[kp], vpc.x * vpc.x + vpc.y * vpc.y, acc);
        // Y_k = wf_k V_k ; conj(Y_p) = conj(wf_p) conj(V_p)
        const float2 a = wf[k], c = wf[kp];
        const float2 yk = make_float2(a.x * vk.x - a.y * vk.y, a.x * vk.y + a.y * vk.x);
        const float2 ypc = make_float2(c.x * vpc.x + c.y * vpc.y, c.x * vpc.y - c.y * vpc.x);
        const float2 ye = make_float2(yk.x + ypc.x, yk.y + ypc.y);
        const float2 d = make_float2(yk.x - ypc.x, yk.y - ypc.y);
        // Yo = d conj(t)
        const float2 yo = make_float2(d.x * cs + d.y * sn, d.y * cs - d.x * sn);
        Zo[k] = make_float2(ye.x - yo.y, ye.y + yo.x);                  // Ye + i Yo
        if (p != k) Zo[p] = make_float2(ye.x + yo.y, -(ye.y - yo.x));   // conj(Ye - i Yo)
    }
    if (!wf) return;
    acc = ofx_block_sum(acc, scratch);
    if (threadIdx.x == 0) chi0[b] = acc;
}

// ---------------------------------------------------------------------------
// search: arg-max of A^2 over the rolled window(s), chi2, low-frequency chi2,
// record write.  One block per trace (qp.OF1x1.calc + get_result_* --
// algorithms.py:336-341, 414-421, 538-558).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void k_search(OfxPlanDev pd, OfxSlotDev sd,
                                               const float* __restrict__ amps,
                                               const float2* __restrict__ vlow, int cap,
                                               const float* __restrict__ chi0v,
                                               const uint8_t* __restrict__ valid,
                                               float* __restrict__ out) {
    __shared__ OfxCand cscratch[RB / OFX_WAVE];
    __shared__ float scratch[RB / OFX_WAVE];
    const size_t b = blockIdx.x;
    float* row = out + b * pd.row;
    if (valid && !valid[b]) {
        for (int q = 0; q < sd.n_search; ++q)
            for (int j = threadIdx.x; j < OFX_SEARCH_FLOATS; j += RB)
                row[sd.search[q].out_off + j] = OFX_SENTINEL;
        return;
    }
    const int N = pd.N, pre = pd.pre;
    const float* a = amps + b * N;
    const float2* V = vlow + b * cap;
    const float chi0 = chi0v[b];

    for (int q = 0; q < sd.n_search; ++q) {
        const OfxSearchDev sq = sd.search[q];
        // a thread visits its rolled bins in increasing order, so "larger A^2, ties to the
        // smaller index" is a strict greater-than; the amplitude is fetched once at the end
        OfxCand best = ofx_cand_none();
        auto scan = [&](int i0, int i1) {
#pragma unroll 4
            for (int i = i0 + threadIdx.x; i < i1; i += RB) {
                int n = i - pre;
                if (n < 0) n += N;
                const float v = a[n];
                const float key = v * v;
                if (key > best.key) {
                    best.key = key;
                    best.idx = i;
                }
            }
        };
        if (sq.outside) {
            scan(0, sq.lo);
            scan(sq.hi, N);
        } else {
            scan(sq.lo, sq.hi);
        }
        if (best.idx != 0x7fffffff) {
            int n = best.idx - pre;
            if (n < 0) n += N;
            best.amp = a[n];
        }
        best = ofx_cand_block_reduce(best, cscratch);
        const int d = best.idx - pre;
        OfxRefined ref;
        ref.amp = best.amp;
        ref.frac = 0.0f;
        if (sq.interp && best.idx > 0 && best.idx < N - 1) {
            // every thread reads the same two neighbours (rolled bins idx -+ 1)
            int nm = best.idx - 1 - pre, np = best.idx + 1 - pre;
            if (nm < 0) nm += N;
            if (np < 0) np += N;
            ref = ofx_interpolate(a[nm], best.amp, a[np], best.idx, N, sd.norm, chi0);
        } else if (sq.interp) {
            ref = ofx_interpolate(0.f, best.amp, 0.f, best.idx, N, sd.norm, chi0);
        }
        float low = 0.0f;
        for (int k = threadIdx.x; k < sq.nlow; k += RB)
            low += ofx_lowchi2_term(k, N, d, ref.amp, V[k], sd.s[k], sd.g[k], ref.frac);
        low = ofx_block_sum(low, scratch);
        if (threadIdx.x == 0)
            ofx_write_search(row, sq, sd, pd.inv_fs, pre, chi0, best, low,
                             sq.interp ? &ref : nullptr);
    }
}

// ---------------------------------------------------------------------------
static int get_fft(ofx_plan* p, int batch, hipStream_t st, OfxFftPlans** out) {
    auto it = p->fft.find(batch);
    if (it == p->fft.end()) {
        {
            const int rc_setup = ofx_rocfft_setup_once();
            if (rc_setup) return rc_setup;
        }
        OfxFftPlans f;
        size_t len = (size_t)p->N / 2;           // packed complex points
        OFX_FFT(rocfft_plan_create(&f.r2c, rocfft_placement_notinplace,
                                   rocfft_transform_type_complex_forward,
                                   rocfft_precision_single, 1, &len, (size_t)batch, nullptr));
        OFX_FFT(rocfft_plan_create(&f.c2r, rocfft_placement_notinplace,
                                   rocfft_transform_type_complex_inverse,
                                   rocfft_precision_single, 1, &len, (size_t)batch, nullptr));
        size_t w1 = 0, w2 = 0;
        OFX_FFT(rocfft_plan_get_work_buffer_size(f.r2c, &w1));
        OFX_FFT(rocfft_plan_get_work_buffer_size(f.c2r, &w2));
        f.work_bytes = w1 > w2 ? w1 : w2;
        if (f.work_bytes) OFX_HIP(hipMalloc(&f.work, f.work_bytes));
        OFX_FFT(rocfft_execution_info_create(&f.info_r2c));
        OFX_FFT(rocfft_execution_info_create(&f.info_c2r));
        if (f.work_bytes) {
            OFX_FFT(rocfft_execution_info_set_work_buffer(f.info_r2c, f.work, f.work_bytes));
            OFX_FFT(rocfft_execution_info_set_work_buffer(f.info_c2r, f.work, f.work_bytes));
        }
        it = p->fft.emplace(batch, f).first;
    }
    OFX_FFT(rocfft_execution_info_set_stream(it->second.info_r2c, st));
    OFX_FFT(rocfft_execution_info_set_stream(it->second.info_c2r, st));
    *out = &it->second;
    return OFX_OK;
}

int ofx_rocfft_release(ofx_plan* p) {
    for (auto& kv : p->fft) {
        OfxFftPlans& f = kv.second;
        if (f.r2c) rocfft_plan_destroy(f.r2c);
        if (f.c2r) rocfft_plan_destroy(f.c2r);
        if (f.info_r2c) rocfft_execution_info_destroy(f.info_r2c);
        if (f.info_c2r) rocfft_execution_info_destroy(f.info_c2r);
        if (f.work) (void)hipFree(f.work);
    }
    p->fft.clear();
    ofx_ldsfft_destroy(p->ldsfft);
    p->ldsfft = nullptr;
    p->ldsfft_tried = false;
    if (p->d_trace) (void)hipFree(p->d_trace);
    if (p->d_spec) (void)hipFree(p->d_spec);
    if (p->d_filt) (void)hipFree(p->d_filt);
    if (p->d_amp) (void)hipFree(p->d_amp);
    if (p->d_chi0) (void)hipFree(p->d_chi0);
    p->d_trace = nullptr;
    p->d_spec = p->d_filt = nullptr;
    p->d_amp = p->d_chi0 = nullptr;
    if (p->d_vlow) (void)hipFree(p->d_vlow);
    p->d_vlow = nullptr;
    p->vlow_cap = 0;
    return OFX_OK;
}

int ofx_rocfft_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                       long long n, float* d_out, hipStream_t st) {
    const int N = p->N, K = p->K, MB = p->max_batch;
    bool any_slot = !p->bands.empty();       // bands need the forward FFT too
    for (int s = 0; s < OFX_MAX_SLOTS; ++s)
        if (p->slot[s].set && !p->slot[s].searches.empty()) any_slot = true;
    const bool need_combine = p->n_channels > 1 || p->n_terms > 1 || p->weight[0] != 1.0;
    const bool need_prep = need_combine || !p->tdwin.empty();

    if (any_slot && !p->d_spec) {
        OFX_HIP(hipMalloc(&p->d_spec, sizeof(float2) * (size_t)MB * K));
        OFX_HIP(hipMalloc(&p->d_filt, sizeof(float2) * (size_t)MB * K));
        OFX_HIP(hipMalloc(&p->d_amp, sizeof(float) * (size_t)MB * N));
        OFX_HIP(hipMalloc(&p->d_chi0, sizeof(float) * (size_t)MB));
    }
    // low bins of the trace spectrum kept for lowchi2 / psd_amp
    int cap = 1;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s)
        if (p->slot[s].set)
            for (const OfxSearchDev& q : p->slot[s].searches) cap = std::max(cap, q.nlow);
    for (const auto& bd : p->bands) cap = std::max(cap, bd.k_hi);
    if (cap > K) cap = K;
    if (any_slot && p->vlow_cap < cap) {
        if (p->d_vlow) (void)hipFree(p->d_vlow);
        p->d_vlow = nullptr;
        p->vlow_cap = 0;
        OFX_HIP(hipMalloc(&p->d_vlow, sizeof(float2) * (size_t)MB * cap));
        p->vlow_cap = cap;
    }
    cap = p->vlow_cap > 0 ? p->vlow_cap : cap;
    if (need_combine && any_slot && !p->d_trace)
        OFX_HIP(hipMalloc(&p->d_trace, sizeof(float) * (size_t)MB * N));

    OfxPlanDev pd;
    ofx_fill_plan_dev(p, &pd);

    for (long long b0 = 0; b0 < n; b0 += MB) {
        const int nb = (int)((n - b0 < MB) ? (n - b0) : MB);
        const float* ev = d_traces + (size_t)b0 * p->n_channels * N;
        const uint8_t* vld = d_valid ? d_valid + b0 : nullptr;
        float* out = d_out + (size_t)b0 * pd.row;
        const float* tr = ev;
        if (need_prep) {
            float* comb = (need_combine && any_slot) ? p->d_trace : nullptr;
            hipLaunchKernelGGL(k_prep, dim3(nb), dim3(RB), 0, st, pd, ev, vld, comb, out);
            if (comb) tr = comb;
        }
        if (!any_slot) continue;
        // transforms: rocFFT, except for non-power-of-two lengths of the form 2^a 3^b 5^c that
        // fit in LDS, where rocFFT takes a multi-kernel path and the LDS transform is faster
        if (!p->ldsfft_tried) {
            p->ldsfft_tried = true;
            const int Mh = N / 2;
            if ((Mh & (Mh - 1)) != 0) {
                const int r = ofx_ldsfft_create(Mh, p->device, &p->ldsfft);
                if (r != OFX_OK && r != OFX_ERR_UNSUPPORTED) return r;
            }
        }
        OfxFftPlans* f = nullptr;
        int rc = p->ldsfft ? OFX_OK : get_fft(p, nb, st, &f);
        if (rc) return rc;
        void* in1[1] = {(void*)tr};
        void* out1[1] = {(void*)p->d_spec};
        size_t tix = 0;
        rc = ofx_time_begin(p, st, &tix);
        if (rc) return rc;
        if (p->ldsfft) {
            rc = ofx_ldsfft_exec(p->ldsfft, true, reinterpret_cast<const float2*>(tr), p->d_spec,
                                 nb, st);
            if (rc) return rc;
        } else {
            OFX_FFT(rocfft_execute(f->r2c, in1, out1, f->info_r2c));
        }
        bool bands_done = p->bands.empty();
        bool any_search = false;
        for (int s = 0; s < OFX_MAX_SLOTS; ++s) {
            if (!p->slot[s].set || p->slot[s].searches.empty()) continue;
            any_search = true;
            OfxSlotDev sd;
            ofx_fill_slot_dev(p, s, &sd);
            hipLaunchKernelGGL(k_mid, dim3(nb), dim3(RB), 0, st, N / 2, N, sd.wf, sd.g, p->d_spec,
                               p->d_filt, p->d_chi0, p->d_vlow, cap);
            if (!bands_done) {
                hipLaunchKernelGGL(k_bands, dim3(nb), dim3(RB), 0, st, pd, p->d_vlow, cap, vld, out);
                bands_done = true;
            }
            if (p->ldsfft) {
                rc = ofx_ldsfft_exec(p->ldsfft, false, p->d_filt,
                                     reinterpret_cast<float2*>(p->d_amp), nb, st);
                if (rc) return rc;
            } else {
                void* in2[1] = {(void*)p->d_filt};
                void* out2[1] = {(void*)p->d_amp};
                OFX_FFT(rocfft_execute(f->c2r, in2, out2, f->info_c2r));
            }
            hipLaunchKernelGGL(k_search, dim3(nb), dim3(RB), 0, st, pd, sd, p->d_amp,
                               p->d_vlow, cap, p->d_chi0, vld, out);
        }
        if (!any_search && !bands_done) {        // psd_amp without a filter slot
            hipLaunchKernelGGL(k_mid, dim3(nb), dim3(RB), 0, st, N / 2, N,
                               (const float2*)nullptr, (const float*)nullptr, p->d_spec,
                               (float2*)nullptr, (float*)nullptr, p->d_vlow, cap);
            hipLaunchKernelGGL(k_bands, dim3(nb), dim3(RB), 0, st, pd, p->d_vlow, cap, vld, out);
        }
        rc = ofx_time_end(p, st, tix);
        if (rc) return rc;
        OFX_HIP(hipGetLastError());
    }
    return OFX_OK;
}
