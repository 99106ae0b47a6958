// ofx_capi.cpp -- the C ABI declared in include/ofx.h (plan management, staging,
// dispatch to the FUSED / ROCFFT engines).  No arithmetic of the hot path
// lives here.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "ofx_common.h"

static thread_local char g_err[1024] = "";

void ofx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ofx_last_error(void) { return g_err; }

extern "C" int ofx_device_info(int device, char* buf, size_t buflen) {
    hipDeviceProp_t prop;
    OFX_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "ofx-0.1;%s;%d;%zu", prop.gcnArchName,
             prop.multiProcessorCount, (size_t)prop.totalGlobalMem);
    return OFX_OK;
}

int ofx_row_floats(const ofx_plan* p) {
    int n = 0;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s)
        if (p->slot[s].set) n += (int)p->slot[s].searches.size() * OFX_SEARCH_FLOATS;
    n += (int)p->tdwin.size() * OFX_TDWIN_FLOATS;
    n += (int)p->bands.size() * OFX_BAND_FLOATS;
    return n;
}

int ofx_rocfft_setup_once() {
    static std::once_flag once;
    static rocfft_status st = rocfft_status_success;
    std::call_once(once, [] { st = rocfft_setup(); });
    OFX_FFT(st);
    return OFX_OK;
}

static void assign_offsets(ofx_plan* p) {
    int off = 0;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s) {
        if (!p->slot[s].set) continue;
        for (auto& q : p->slot[s].searches) {
            q.out_off = off;
            off += OFX_SEARCH_FLOATS;
        }
    }
    for (auto& w : p->tdwin) {
        w.out_off = off;
        off += OFX_TDWIN_FLOATS;
    }
    for (auto& bd : p->bands) {
        bd.out_off = off;
        off += OFX_BAND_FLOATS;
    }
}

void ofx_fill_plan_dev(const ofx_plan* p, OfxPlanDev* d) {
    memset(d, 0, sizeof(*d));
    d->N = p->N;
    d->K = p->K;
    d->pre = p->pre;
    d->fs = (float)p->fs;
    d->inv_fs = (float)(1.0 / p->fs);
    d->row = ofx_row_floats(p);
    d->n_channels = p->n_channels;
    d->n_terms = p->n_terms;
    for (int j = 0; j < OFX_MAX_TERMS; ++j) {
        d->chan[j] = p->chan[j];
        d->weight[j] = (float)p->weight[j];
    }
    d->n_tdwin = (int)p->tdwin.size();
    for (int w = 0; w < d->n_tdwin; ++w) d->tdw[w] = p->tdwin[w];
    d->n_bands = (int)p->bands.size();
    for (int i = 0; i < d->n_bands; ++i) d->band[i] = p->bands[i];
}

void ofx_fill_slot_dev(const ofx_plan* p, int slot, OfxSlotDev* d) {
    const OfxSlotHost& h = p->slot[slot];
    memset(d, 0, sizeof(*d));
    d->wf = h.d_wf;
    d->g = h.d_g;
    d->s = h.d_s;
    d->pq = h.d_pq;
    d->norm = (float)h.norm;
    d->tres_sum = (float)h.tres_sum;
    d->ampres = (float)(1.0 / std::sqrt(h.norm));
    d->n_search = (int)h.searches.size();
    for (int q = 0; q < d->n_search; ++q) d->search[q] = h.searches[q];
}

extern "C" int ofx_plan_create(ofx_plan** out, int n_samples, int n_pretrigger,
                               double fs, int max_batch, int device, int engine) {
    if (!out || n_samples < 8 || n_pretrigger < 0 ||
        n_pretrigger >= n_samples || !(fs > 0) || max_batch < 1) {
        ofx_set_error("ofx_plan_create: bad argument (n_samples=%d must be >= 8, "
                      "0 <= n_pretrigger=%d < n_samples, fs=%g > 0, max_batch=%d >= 1)",
                      n_samples, n_pretrigger, fs, max_batch);
        return OFX_ERR_ARG;
    }
    if (engine == OFX_ENGINE_FUSED && !ofx_fused_supported(n_samples)) {
        ofx_set_error("ofx_plan_create: FUSED engine does not support n_samples=%d",
                      n_samples);
        return OFX_ERR_UNSUPPORTED;
    }
    if (engine == OFX_ENGINE_LDS && !ofx_lds_supported(n_samples)) {
        ofx_set_error("ofx_plan_create: LDS engine does not support n_samples=%d (N/2 must be "
                      "2^a 3^b 5^c and N <= 34816)", n_samples);
        return OFX_ERR_UNSUPPORTED;
    }
    OFX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    OFX_HIP(hipGetDeviceProperties(&prop, device));
    ofx_plan* p = new ofx_plan();
    p->N = n_samples;
    p->K = n_samples / 2 + 1;
    p->pre = n_pretrigger;
    p->fs = fs;
    p->max_batch = max_batch;
    p->device = device;
    p->cu_count = prop.multiProcessorCount;
    if (engine == OFX_ENGINE_AUTO) {
        p->engine_auto = true;
        // measured: the LDS engine wins wherever rocFFT needs several passes
        // (lengths that are not powers of two) and for 4096..8192 samples; rocFFT's
        // single-kernel power-of-two transforms win below and above that
        const bool pow2 = (n_samples & (n_samples - 1)) == 0;
        const bool lds_wins = ofx_lds_supported(n_samples) &&
                              !(pow2 && (n_samples <= 2048 || n_samples >= 16384));
        engine = ofx_fused_supported(n_samples) ? OFX_ENGINE_FUSED
                 : lds_wins                     ? OFX_ENGINE_LDS
                                                : OFX_ENGINE_ROCFFT;
    }
    p->engine = engine;
    p->chan[0] = 0;
    p->weight[0] = 1.0;
    *out = p;
    return OFX_OK;
}

static void free_slot(OfxSlotHost& h) {
    if (h.d_wf) (void)hipFree(h.d_wf);
    if (h.d_g) (void)hipFree(h.d_g);
    if (h.d_s) (void)hipFree(h.d_s);
    if (h.d_pq) (void)hipFree(h.d_pq);
    h = OfxSlotHost();
}

extern "C" int ofx_plan_reset(ofx_plan* p) {
    if (!p) return OFX_ERR_ARG;
    (void)hipSetDevice(p->device);
    for (int s = 0; s < OFX_MAX_SLOTS; ++s) free_slot(p->slot[s]);
    ++p->filter_stamp;
    p->tdwin.clear();
    p->bands.clear();
    p->n_channels = 1;
    p->n_terms = 1;
    p->chan[0] = 0;
    p->weight[0] = 1.0;
    return OFX_OK;
}

extern "C" int ofx_plan_destroy(ofx_plan* p) {
    if (!p) return OFX_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    ofx_plan_reset(p);
    ofx_rocfft_release(p);
    ofx_fused_release(p);
    ofx_lds_release(p);
    if (p->d_adc) (void)hipFree(p->d_adc);
    if (p->d_trig) (void)hipFree(p->d_trig);
    if (p->d_stage_in) (void)hipFree(p->d_stage_in);
    if (p->d_stage_valid) (void)hipFree(p->d_stage_valid);
    if (p->d_stage_out) (void)hipFree(p->d_stage_out);
    for (auto& e : p->ev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete p;
    return OFX_OK;
}

extern "C" int ofx_plan_engine(const ofx_plan* p) { return p ? p->engine : -1; }

extern "C" int ofx_plan_set_filter(ofx_plan* p, int slot, const double* wf,
                                   const double* g, const double* s, double norm,
                                   double tres_sum) {
    if (!p || slot < 0 || slot >= OFX_MAX_SLOTS || !wf || !g || !s || !(norm > 0)) {
        ofx_set_error("ofx_plan_set_filter: bad argument (slot=%d, norm=%g)", slot, norm);
        return OFX_ERR_ARG;
    }
    if (p->N & 1) {
        ofx_set_error("ofx_plan_set_filter: odd trace lengths (%d) carry time-domain "
                      "windows only", p->N);
        return OFX_ERR_ARG;
    }
    OFX_HIP(hipSetDevice(p->device));
    // convert and validate first: a rejected table leaves the slot as it was
    const int K = p->K;
    std::vector<float2> wf32(K), s32(K);
    std::vector<float> g32(K);
    for (int k = 0; k < K; ++k) {
        wf32[k] = make_float2((float)wf[2 * k], (float)wf[2 * k + 1]);
        s32[k] = make_float2((float)s[2 * k], (float)s[2 * k + 1]);
        g32[k] = (float)g[k];
        if (!std::isfinite(wf32[k].x) || !std::isfinite(wf32[k].y) ||
            !std::isfinite(s32[k].x) || !std::isfinite(s32[k].y) ||
            !std::isfinite(g32[k]) || g32[k] < 0) {
            ofx_set_error("ofx_plan_set_filter: non-finite / negative table entry at "
                          "bin %d (does the filter overflow fp32?)", k);
            return OFX_ERR_ARG;
        }
    }
    OfxSlotHost& h = p->slot[slot];
    std::vector<OfxSearchDev> keep = h.searches;
    // the device tables of the old filter may still be read by a launch in flight
    OFX_HIP(hipDeviceSynchronize());
    free_slot(h);
    h.searches = keep;
    ++p->filter_stamp;              // whatever happens below, the slot tables are stale
    assign_offsets(p);              // ... and the slot is unset until the upload has succeeded
    OFX_HIP(hipMalloc(&h.d_wf, sizeof(float2) * K));
    OFX_HIP(hipMalloc(&h.d_g, sizeof(float) * K));
    OFX_HIP(hipMalloc(&h.d_s, sizeof(float2) * K));
    OFX_HIP(hipMemcpy(h.d_wf, wf32.data(), sizeof(float2) * K, hipMemcpyHostToDevice));
    OFX_HIP(hipMemcpy(h.d_g, g32.data(), sizeof(float) * K, hipMemcpyHostToDevice));
    OFX_HIP(hipMemcpy(h.d_s, s32.data(), sizeof(float2) * K, hipMemcpyHostToDevice));
    h.norm = norm;
    h.tres_sum = tres_sum;
    h.g_host.assign(g, g + K);
    h.wf_host.assign(wf, wf + 2 * (size_t)K);
    h.set = true;
    if (p->engine == OFX_ENGINE_FUSED) {
        int rc = ofx_fused_prepare_slot(p, slot, wf);
        if (rc != OFX_OK) return rc;
    }
    assign_offsets(p);
    return OFX_OK;
}

extern "C" int ofx_plan_add_search(ofx_plan* p, int slot, int kind, int lo, int hi,
                                   int outside, double fcut) {
    if (!p || slot < 0 || slot >= OFX_MAX_SLOTS || !p->slot[slot].set) {
        ofx_set_error("ofx_plan_add_search: slot %d has no filter", slot);
        return -OFX_ERR_STATE;
    }
    if (kind != OFX_SEARCH_NODELAY && kind != OFX_SEARCH_DELAY &&
        kind != OFX_SEARCH_DELAY_INTERP) {
        ofx_set_error("ofx_plan_add_search: unknown kind %d", kind);
        return -OFX_ERR_ARG;
    }
    OfxSlotHost& h = p->slot[slot];
    if ((int)h.searches.size() >= OFX_MAX_SEARCHES) {
        ofx_set_error("ofx_plan_add_search: more than %d searches on slot %d",
                      OFX_MAX_SEARCHES, slot);
        return -OFX_ERR_ARG;
    }
    OfxSearchDev q;
    memset(&q, 0, sizeof(q));
    q.interp = (kind == OFX_SEARCH_DELAY_INTERP) ? 1 : 0;
    if (q.interp) kind = OFX_SEARCH_DELAY;
    q.kind = kind;
    if (kind == OFX_SEARCH_NODELAY) {
        q.lo = p->pre;
        q.hi = p->pre + 1;
        q.outside = 0;
    } else {
        if (lo < 0) lo = 0;
        if (hi > p->N) hi = p->N;
        q.lo = lo;
        q.hi = hi;
        q.outside = outside ? 1 : 0;
        int count = q.outside ? (p->N - (hi > lo ? hi - lo : 0)) : (hi - lo);
        if (count <= 0) {
            ofx_set_error("ofx_plan_add_search: empty search range [%d,%d) outside=%d",
                          lo, hi, outside);
            return -OFX_ERR_ARG;
        }
        if (q.outside && hi <= lo) { q.lo = 0; q.hi = 0; }
    }
    // bins with |f_k| <= fcut, f_k computed as numpy.fft.fftfreq does
    const double val = 1.0 / (p->N * (1.0 / p->fs));
    int nlow = 0;
    for (int k = 0; k < p->K; ++k)
        if (k * val <= fcut) nlow = k + 1; else break;
    q.nlow = nlow;
    h.searches.push_back(q);
    ++p->filter_stamp;
    assign_offsets(p);
    return (int)h.searches.size() - 1;
}

extern "C" int ofx_plan_add_tdwindow(ofx_plan* p, int lo, int hi) {
    if (!p) return -OFX_ERR_ARG;
    if ((int)p->tdwin.size() >= OFX_MAX_TDWIN) {
        ofx_set_error("ofx_plan_add_tdwindow: more than %d windows", OFX_MAX_TDWIN);
        return -OFX_ERR_ARG;
    }
    // numpy slice semantics on trace[lo:hi]
    if (lo < 0) lo = 0;
    if (hi > p->N) hi = p->N;
    if (hi <= lo) {
        ofx_set_error("ofx_plan_add_tdwindow: empty slice [%d:%d]", lo, hi);
        return -OFX_ERR_ARG;
    }
    OfxTdWinDev w;
    w.lo = lo;
    w.hi = hi;
    w.out_off = 0;
    p->tdwin.push_back(w);
    assign_offsets(p);
    return (int)p->tdwin.size() - 1;
}

extern "C" int ofx_plan_add_band(ofx_plan* p, int k_lo, int k_hi) {
    if (!p) return -OFX_ERR_ARG;
    if ((int)p->bands.size() >= OFX_MAX_BANDS) {
        ofx_set_error("ofx_plan_add_band: more than %d bands", OFX_MAX_BANDS);
        return -OFX_ERR_ARG;
    }
    if (k_lo < 1 || k_hi <= k_lo || k_hi > p->K || (p->N & 1)) {
        ofx_set_error("ofx_plan_add_band: need 1 <= k_lo=%d < k_hi=%d <= %d (even trace "
                      "length)", k_lo, k_hi, p->K);
        return -OFX_ERR_ARG;
    }
    OfxBandDev bd;
    bd.k_lo = k_lo;
    bd.k_hi = k_hi;
    bd.out_off = 0;
    p->bands.push_back(bd);
    assign_offsets(p);
    return (int)p->bands.size() - 1;
}

extern "C" int ofx_plan_set_channels(ofx_plan* p, int n_channels, int n_terms,
                                     const int* chan_index, const double* weight) {
    if (!p || n_channels < 1 || n_terms < 1 || n_terms > OFX_MAX_TERMS || !chan_index) {
        ofx_set_error("ofx_plan_set_channels: bad argument");
        return OFX_ERR_ARG;
    }
    for (int j = 0; j < n_terms; ++j) {
        if (chan_index[j] < 0 || chan_index[j] >= n_channels) {
            ofx_set_error("ofx_plan_set_channels: channel index %d out of range",
                          chan_index[j]);
            return OFX_ERR_ARG;
        }
    }
    p->n_channels = n_channels;
    p->n_terms = n_terms;
    for (int j = 0; j < n_terms; ++j) {
        p->chan[j] = chan_index[j];
        p->weight[j] = weight ? weight[j] : 1.0;
    }
    return OFX_OK;
}

extern "C" int ofx_plan_row_floats(const ofx_plan* p) { return p ? ofx_row_floats(p) : -1; }

extern "C" int ofx_plan_search_offset(const ofx_plan* p, int slot, int search) {
    if (!p || slot < 0 || slot >= OFX_MAX_SLOTS || !p->slot[slot].set || search < 0 ||
        search >= (int)p->slot[slot].searches.size())
        return -1;
    return p->slot[slot].searches[search].out_off;
}

extern "C" int ofx_plan_tdwindow_offset(const ofx_plan* p, int w) {
    if (!p || w < 0 || w >= (int)p->tdwin.size()) return -1;
    return p->tdwin[w].out_off;
}

extern "C" int ofx_plan_band_offset(const ofx_plan* p, int i) {
    if (!p || i < 0 || i >= (int)p->bands.size()) return -1;
    return p->bands[i].out_off;
}

// ------------------------------------------------------------------- timing
extern "C" int ofx_plan_enable_timing(ofx_plan* p, int enable) {
    if (!p) return OFX_ERR_ARG;
    p->timing = enable != 0;
    return OFX_OK;
}

int ofx_time_begin(ofx_plan* p, hipStream_t st, size_t* idx) {
    if (!p->timing) return OFX_OK;
    if (p->ev_used == p->ev.size()) {
        hipEvent_t a, b;
        OFX_HIP(hipEventCreate(&a));
        OFX_HIP(hipEventCreate(&b));
        p->ev.emplace_back(a, b);
    }
    *idx = p->ev_used++;
    OFX_HIP(hipEventRecord(p->ev[*idx].first, st));
    return OFX_OK;
}

int ofx_time_end(ofx_plan* p, hipStream_t st, size_t idx) {
    if (!p->timing) return OFX_OK;
    OFX_HIP(hipEventRecord(p->ev[idx].second, st));
    return OFX_OK;
}

extern "C" int ofx_plan_kernel_time(ofx_plan* p, double* avg_ms, long long* n) {
    if (!p) return OFX_ERR_ARG;
    OFX_HIP(hipSetDevice(p->device));
    for (size_t i = 0; i < p->ev_used; ++i) {
        OFX_HIP(hipEventSynchronize(p->ev[i].second));
        float ms = 0;
        OFX_HIP(hipEventElapsedTime(&ms, p->ev[i].first, p->ev[i].second));
        p->t_acc_ms += ms;
        p->t_launches += 1;
    }
    p->ev_used = 0;
    if (avg_ms) *avg_ms = p->t_launches ? p->t_acc_ms / (double)p->t_launches : 0.0;
    if (n) *n = p->t_launches;
    p->t_acc_ms = 0;
    p->t_launches = 0;
    return OFX_OK;
}

// ------------------------------------------------------------------ process
static int ensure(float** buf, size_t* have, size_t want) {
    if (*have >= want) return OFX_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    OFX_HIP(hipMalloc(buf, want * sizeof(float)));
    *have = want;
    return OFX_OK;
}

// One engine call on device buffers.  AUTO plans fall back to the general engine when the
// chosen fused engine reports that it cannot carry the plan (e.g. a lowchi2 cutoff or a
// psd_amp band beyond the bins it stashes).
static int engine_process(ofx_plan* p, const float* d_in, const uint8_t* d_valid, long long n,
                          float* d_out, hipStream_t st) {
    int rc;
    if (p->engine == OFX_ENGINE_FUSED)
        rc = ofx_fused_process(p, d_in, d_valid, n, d_out, st);
    else if (p->engine == OFX_ENGINE_LDS)
        rc = ofx_lds_process(p, d_in, d_valid, n, d_out, st);
    else
        return ofx_rocfft_process(p, d_in, d_valid, n, d_out, st);
    if (rc == OFX_ERR_UNSUPPORTED && p->engine_auto) {
        // (a FUSED plan of a length the LDS engine carries faster than the rocFFT pipeline, e.g.
        // 25000 or 4096 samples, tries that one first)
        const bool pow2 = (p->N & (p->N - 1)) == 0;
        const bool lds_wins = ofx_lds_supported(p->N) && !(pow2 && (p->N <= 2048 || p->N >= 16384));
        if (p->engine == OFX_ENGINE_FUSED && lds_wins) {
            rc = ofx_lds_process(p, d_in, d_valid, n, d_out, st);
            if (rc != OFX_ERR_UNSUPPORTED) return rc;
        }
        rc = ofx_rocfft_process(p, d_in, d_valid, n, d_out, st);
    }
    return rc;
}

extern "C" int ofx_process(ofx_plan* p, const float* traces, const uint8_t* valid,
                           long long n, int traces_mem, float* out, int out_mem,
                           void* stream) {
    if (!p || n < 0 || (n > 0 && (!traces || !out))) {
        ofx_set_error("ofx_process: bad argument");
        return OFX_ERR_ARG;
    }
    const int row = ofx_row_floats(p);
    if (row == 0) {
        ofx_set_error("ofx_process: plan has no searches, time-domain windows or bands");
        return OFX_ERR_STATE;
    }
    if (n == 0) return OFX_OK;
    OFX_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t ev_floats = (size_t)p->n_channels * p->N;

    if (traces_mem == OFX_MEM_DEVICE && out_mem == OFX_MEM_DEVICE) {
        return engine_process(p, traces, valid, n, out, st);
    }

    // host buffers on either side: stage chunk by chunk (PCIe-inclusive path)
    const long long chunk = p->max_batch;
    for (long long b0 = 0; b0 < n; b0 += chunk) {
        const long long nb = (n - b0 < chunk) ? (n - b0) : chunk;
        const float* d_in = traces + (size_t)b0 * ev_floats;
        const uint8_t* d_valid = valid ? valid + b0 : nullptr;
        float* d_out = out + (size_t)b0 * row;
        if (traces_mem == OFX_MEM_HOST) {
            int rc = ensure(&p->d_stage_in, &p->stage_in_floats, (size_t)chunk * ev_floats);
            if (rc) return rc;
            OFX_HIP(hipMemcpyAsync(p->d_stage_in, traces + (size_t)b0 * ev_floats,
                                   (size_t)nb * ev_floats * sizeof(float),
                                   hipMemcpyHostToDevice, st));
            d_in = p->d_stage_in;
            if (valid) {
                if (!p->d_stage_valid || p->stage_valid_elems < (size_t)chunk) {
                    if (p->d_stage_valid) (void)hipFree(p->d_stage_valid);
                    p->d_stage_valid = nullptr;
                    OFX_HIP(hipMalloc(&p->d_stage_valid, (size_t)chunk));
                    p->stage_valid_elems = (size_t)chunk;
                }
                OFX_HIP(hipMemcpyAsync(p->d_stage_valid, valid + b0, (size_t)nb,
                                       hipMemcpyHostToDevice, st));
                d_valid = p->d_stage_valid;
            }
        }
        if (out_mem == OFX_MEM_HOST) {
            int rc = ensure(&p->d_stage_out, &p->stage_out_floats, (size_t)chunk * row);
            if (rc) return rc;
            d_out = p->d_stage_out;
        }
        int rc = engine_process(p, d_in, d_valid, nb, d_out, st);
        if (rc) return rc;
        if (out_mem == OFX_MEM_HOST) {
            OFX_HIP(hipMemcpyAsync(out + (size_t)b0 * row, d_out,
                                   (size_t)nb * row * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        }
        OFX_HIP(hipStreamSynchronize(st));
    }
    return OFX_OK;
}

// ------------------------------------------------------------ process (ADC)
extern "C" int ofx_process_adc(ofx_plan* p, const int16_t* adc, long long n_stream, int adc_mem,
                               const long long* trigger_index, long long n, const double* scale,
                               const double* offset, float* out, int out_mem, void* stream) {
    if (!p || n < 0 || n_stream < 0 || (n > 0 && (!adc || !trigger_index || !out)) || !scale ||
        !offset) {
        ofx_set_error("ofx_process_adc: bad argument");
        return OFX_ERR_ARG;
    }
    const int row = ofx_row_floats(p);
    if (row == 0) {
        ofx_set_error("ofx_process_adc: plan has no searches, time-domain windows or bands");
        return OFX_ERR_STATE;
    }
    if (n == 0) return OFX_OK;
    OFX_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const int C = p->n_channels;
    const size_t ev_floats = (size_t)C * p->N;

    // the stream(s): staged once, whatever the number of windows
    const int16_t* d_adc = adc;
    if (adc_mem == OFX_MEM_HOST) {
        const size_t want = (size_t)C * (size_t)n_stream;
        if (p->adc_elems < want) {
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
        if (p->d_trig) (void)hipFree(p->d_trig);
        p->d_trig = nullptr;
        p->trig_elems = 0;
        OFX_HIP(hipMalloc(&p->d_trig, (size_t)n * sizeof(long long)));
        p->trig_elems = (size_t)n;
    }
    OFX_HIP(hipMemcpyAsync(p->d_trig, trigger_index, (size_t)n * sizeof(long long),
                           hipMemcpyHostToDevice, st));
    std::vector<float> sc(C), of(C);
    for (int c = 0; c < C; ++c) {
        sc[c] = (float)scale[c];
        of[c] = (float)offset[c];
    }

    long long chunk = p->max_batch;
    if (chunk > 32768) chunk = 32768;
    int rc = ensure(&p->d_stage_in, &p->stage_in_floats, (size_t)chunk * ev_floats);
    if (rc) return rc;
    if (!p->d_stage_valid || p->stage_valid_elems < (size_t)chunk) {
        if (p->d_stage_valid) (void)hipFree(p->d_stage_valid);
        p->d_stage_valid = nullptr;
        OFX_HIP(hipMalloc(&p->d_stage_valid, (size_t)chunk));
        p->stage_valid_elems = (size_t)chunk;
    }
    if (out_mem == OFX_MEM_HOST) {
        rc = ensure(&p->d_stage_out, &p->stage_out_floats, (size_t)chunk * row);
        if (rc) return rc;
    }
    for (long long b0 = 0; b0 < n; b0 += chunk) {
        const long long nb = (n - b0 < chunk) ? (n - b0) : chunk;
        rc = ofx_cut_launch(d_adc, n_stream, C, p->N, p->pre, p->d_trig + b0, nb, sc.data(),
                            of.data(), p->d_stage_in, p->d_stage_valid, st);
        if (rc) return rc;
        float* d_out = (out_mem == OFX_MEM_HOST) ? p->d_stage_out : out + (size_t)b0 * row;
        rc = engine_process(p, p->d_stage_in, p->d_stage_valid, nb, d_out, st);
        if (rc) return rc;
        if (out_mem == OFX_MEM_HOST)
            OFX_HIP(hipMemcpyAsync(out + (size_t)b0 * row, d_out, (size_t)nb * row * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        OFX_HIP(hipStreamSynchronize(st));      // the staging buffers are reused by the next chunk
    }
    return OFX_OK;
}
