// ofx_fused_host.h -- host side of a launch of the register-resident kernels, shared by ofx_fused.hip
// (32768 samples) and ofx_fused25.hip (25000 / 20000 / 12500): per launch the plan is turned into the
// kernel's arguments -- window rows classified, one slot record per filter slot with searches (tables,
// row mask of its windowed fits), the feature bits that pick the kernel instantiation, the slot table
// uploaded for several slots -- and handed to the file's own dispatcher.
//
// G (geometry of the kernel):  N samples, ROWS lags per register row, NROWS rows, LDS_BINS low bins
// kept in LDS, MAX_BINS with the global stash (== LDS_BINS without one), MIDG_OFF offset (float4 units)
// of the g table behind the W table of a slot, Tabs / SlotArg the kernel's table structs.
// launch(multi, feat, pd, sd, tabs, d_slots, nslots, nstash) -> status.
#pragma once

template <class G>
static void fused_classify_windows(OfxPlanDev& pd) {
    for (int w = 0; w < pd.n_tdwin; ++w) {          // row classification of the window sums
        pd.tdw[w].full = pd.tdw[w].edge = 0;
        for (int n1 = 0; n1 < G::NROWS; ++n1) {
            const int r0 = G::ROWS * n1, lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
            if (r0 + G::ROWS <= lo || r0 >= hi) continue;
            if (lo <= r0 && r0 + G::ROWS <= hi) pd.tdw[w].full |= 1u << n1;
            else pd.tdw[w].edge |= 1u << n1;
        }
    }
}

// register rows of the lags n = (i - pre) mod N the rolled indices [i0, i1) visit
template <class G>
static unsigned fused_rows_of(int i0, int i1, int pre) {
    unsigned mask = 0;
    auto lag = [&](int i) {
        int nl = (i - pre) % G::N;
        return nl < 0 ? nl + G::N : nl;
    };
    for (int i = i0; i < i1;) {
        const int nl = lag(i);
        mask |= 1u << (nl / G::ROWS);
        i += G::ROWS - nl % G::ROWS;                 // first lag of the next row
    }
    if (i1 > i0) mask |= 1u << (lag(i1 - 1) / G::ROWS);
    return mask;
}

template <class G, class Launch>
static int fused_process_plan(ofx_plan* p, const typename G::Tabs& common, hipStream_t st, Launch launch) {
    using SlotArg = typename G::SlotArg;
    OfxPlanDev pd;
    ofx_fill_plan_dev(p, &pd);
    fused_classify_windows<G>(pd);
    std::vector<SlotArg> args;
    int feat = 0;
    int nstash = 0;                 // bins of 2 X_k the tail reads back
    for (int s = 0; s < OFX_MAX_SLOTS; ++s) {
        if (!p->slot[s].set || p->slot[s].searches.empty()) continue;
        SlotArg a;
        memset(&a, 0, sizeof(a));
        ofx_fill_slot_dev(p, s, &a.sd);
        a.tabs = common;
        a.tabs.midW = p->slot[s].d_pq;
        a.tabs.midG = reinterpret_cast<const float2*>(p->slot[s].d_pq + G::MIDG_OFF);
        a.tabs.wq = make_float2(p->slot[s].wq_x, p->slot[s].wq_y);
        a.tabs.gq = p->slot[s].gq;
        for (int q = 0; q < a.sd.n_search; ++q) {
            const OfxSearchDev& sq = a.sd.search[q];
            const bool full = sq.lo == 0 && sq.hi == p->N && !sq.outside;
            if (sq.kind == OFX_SEARCH_DELAY && (sq.interp || !full)) feat |= 1;
            if (sq.kind == OFX_SEARCH_DELAY && !full) {
                if (sq.outside)
                    a.tabs.rowmask |= fused_rows_of<G>(0, sq.lo, p->pre) | fused_rows_of<G>(sq.hi, G::N, p->pre);
                else
                    a.tabs.rowmask |= fused_rows_of<G>(sq.lo, sq.hi, p->pre);
            }
            if (sq.nlow > G::MAX_BINS) {
                ofx_set_error("FUSED engine (%d samples): lowchi2_fcutoff covers %d bins (> %d)", G::N,
                              sq.nlow, G::MAX_BINS);
                return OFX_ERR_UNSUPPORTED;
            }
            if (sq.nlow > nstash) nstash = sq.nlow;
        }
        args.push_back(a);
    }
    const int nslots = (int)args.size();
    if (pd.n_bands > 0) {
        if (nslots == 0) {
            ofx_set_error("FUSED engine: psd_amp bands need at least one filter slot with a "
                          "search on the plan (use the ROCFFT engine otherwise)");
            return OFX_ERR_UNSUPPORTED;
        }
        for (int i = 0; i < pd.n_bands; ++i) {
            if (pd.band[i].k_hi > G::MAX_BINS) {
                ofx_set_error("FUSED engine (%d samples): band [%d,%d) exceeds the %d stashed bins", G::N,
                              pd.band[i].k_lo, pd.band[i].k_hi, G::MAX_BINS);
                return OFX_ERR_UNSUPPORTED;
            }
            if (pd.band[i].k_hi > nstash) nstash = pd.band[i].k_hi;
        }
    }
    if (pd.n_tdwin > 0) feat |= 2;
    if (nstash > G::LDS_BINS) feat |= 8;
    if (p->n_channels > 1 || p->n_terms > 1 || p->weight[0] != 1.0) feat |= 4;

    if (nslots <= 1) {
        OfxSlotDev sd;
        memset(&sd, 0, sizeof(sd));
        typename G::Tabs tabs = common;
        if (nslots == 1) {
            sd = args[0].sd;
            tabs = args[0].tabs;
        }
        return launch(false, feat, pd, sd, tabs, static_cast<const SlotArg*>(nullptr), nslots, nstash);
    }
    // several slots: upload the slot table (from a buffer owned by the plan: the copy is
    // asynchronous on the stream)
    const size_t bytes = sizeof(SlotArg) * (size_t)nslots;
    if (!p->d_fused_slots) OFX_HIP(hipMalloc(&p->d_fused_slots, sizeof(SlotArg) * OFX_MAX_SLOTS));
    if (p->fused_slot_stamp != p->filter_stamp) {
        OFX_HIP(hipStreamSynchronize(st));      // an earlier launch may still read the table
        p->h_slot_args.assign(reinterpret_cast<const unsigned char*>(args.data()),
                              reinterpret_cast<const unsigned char*>(args.data()) + bytes);
        OFX_HIP(hipMemcpyAsync(p->d_fused_slots, p->h_slot_args.data(), bytes, hipMemcpyHostToDevice,
                               st));
        p->fused_slot_stamp = p->filter_stamp;
    }
    OfxSlotDev sd0;
    memset(&sd0, 0, sizeof(sd0));
    return launch(true, feat, pd, sd0, common, reinterpret_cast<const SlotArg*>(p->d_fused_slots),
                  nslots, nstash);
}


// Per-workgroup scratch of the several-slots kernels (the parked spectrum), sized for the full grid and
// allocated once per plan.
[[maybe_unused]] static int fused_ensure_spec(ofx_plan* p, size_t need) {
    if (p->fused_spec_bytes < need) {
        if (p->d_fused_spec) (void)hipFree(p->d_fused_spec);
        p->d_fused_spec = nullptr;
        p->fused_spec_bytes = 0;
        OFX_HIP(hipMalloc(&p->d_fused_spec, need));
        p->fused_spec_bytes = need;
    }
    return OFX_OK;
}

// The twiddle tables of a plan on the device (t1: stage-1 anchors, t2: stage-2 twiddles + bases).
static int fused_upload_tables(ofx_plan* p, const std::vector<float2>& t1, const std::vector<float2>& t2) {
    OFX_HIP(hipMalloc(&p->d_tw1, sizeof(float2) * t1.size()));
    OFX_HIP(hipMalloc(&p->d_tw2, sizeof(float2) * t2.size()));
    OFX_HIP(hipMemcpy(p->d_tw1, t1.data(), sizeof(float2) * t1.size(), hipMemcpyHostToDevice));
    OFX_HIP(hipMemcpy(p->d_tw2, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice));
    return OFX_OK;
}

// Middle-step tables of one filter slot from the fp64 one-sided filter, the same recipe in every
// register-resident kernel: thread (lane) v, pair slot j holds the bin k = bin_of(v, j) and its partner
// M - k (k = 0 pairs DC with Nyquist).
//   d_pq (float4 units): [0 .. nslot * vpad)        midW (W_k / 2, conj(W_p) / 2)   [slot j][v]
//                        [nslot * vpad .. + half)   midG (g_k', g_p') as float2      [slot j][v]
//                        last entry                 (W_{M/2}.x, W_{M/2}.y, g_{M/2}, 0): the self-paired bin
template <class BinOf>
static void fused_fill_slot_tables(float4* tw, float2* tg, const double* wf, const std::vector<double>& g, int M,
                                   int nthreads, int vpad, int nslot, BinOf bin_of) {
    auto W = [&](int k, double& re, double& im) { re = wf[2 * k]; im = wf[2 * k + 1]; };
    for (int v = 0; v < nthreads; ++v) {
        for (int j = 0; j < nslot; ++j) {
            const int k = bin_of(v, j);
            const int pidx = (M - k) % M;
            double wkr, wki, wpr, wpi, gk, gp;
            if (k == 0) {                   // slot (DC, Nyquist): "p" is the Nyquist bin N/2 = M
                W(0, wkr, wki);
                W(M, wpr, wpi);
                gk = g[0] / 4.0;
                gp = g[M] / 4.0;
            } else {
                W(k, wkr, wki);
                W(pidx, wpr, wpi);
                gk = g[k] / 2.0;
                gp = g[pidx] / 2.0;
            }
            tw[j * vpad + v] = make_float4((float)(wkr / 2.0), (float)(wki / 2.0), (float)(wpr / 2.0),
                                           (float)(-wpi / 2.0));
            tg[j * vpad + v] = make_float2((float)gk, (float)gp);
        }
    }
}
// ... the table of the self-paired bin M / 2 behind them, and the upload
static int fused_finish_slot_tables(ofx_plan* p, int slot, const double* wf, std::vector<float4>& tab, int M) {
    OfxSlotHost& h = p->slot[slot];
    const std::vector<double>& g = h.g_host;
    tab.back() = make_float4((float)wf[2 * (M / 2)], (float)wf[2 * (M / 2) + 1], (float)g[M / 2], 0.0f);
    h.wq_x = tab.back().x;
    h.wq_y = tab.back().y;
    h.gq = tab.back().z;
    OFX_HIP(hipMalloc(&h.d_pq, sizeof(float4) * tab.size()));
    OFX_HIP(hipMemcpy(h.d_pq, tab.data(), sizeof(float4) * tab.size(), hipMemcpyHostToDevice));
    return OFX_OK;
}
template <class BinOf>
static int fused_build_slot_tables(ofx_plan* p, int slot, const double* wf, int M, int nthreads, int vpad,
                                   int nslot, BinOf bin_of) {
    const int NW = nslot * vpad, NG = nslot * vpad / 2;
    std::vector<float4> tab(NW + NG + 1, make_float4(0.f, 0.f, 0.f, 0.f));
    fused_fill_slot_tables(tab.data(), reinterpret_cast<float2*>(tab.data() + NW), wf, p->slot[slot].g_host, M,
                           nthreads, vpad, nslot, bin_of);
    return fused_finish_slot_tables(p, slot, wf, tab, M);
}

// The transform of a kernel on its own (rows of M complex points; the N x M engine runs on these): a handle
// with the twiddle tables, and the launch of the forward / inverse kernel.
struct FusedRegFft {
    float2* d_t1 = nullptr;
    float2* d_t2 = nullptr;
    int cu_count = 256;
};
template <class Tables>
[[maybe_unused]] static int fused_fft_create(int device, void** out, Tables tables) {
    ofx_plan tmp;                       // only its table pointers are used
    int rc = tables(&tmp);
    if (rc) return rc;
    FusedRegFft* f = new FusedRegFft();
    f->d_t1 = tmp.d_tw1;
    f->d_t2 = tmp.d_tw2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) f->cu_count = prop.multiProcessorCount;
    *out = f;
    return OFX_OK;
}
[[maybe_unused]] static void fused_fft_destroy(void* h) {
    FusedRegFft* f = static_cast<FusedRegFft*>(h);
    if (!f) return;
    if (f->d_t1) (void)hipFree(f->d_t1);
    if (f->d_t2) (void)hipFree(f->d_t2);
    delete f;
}
template <auto KFWD, auto KINV>
[[maybe_unused]] static int fused_fft_exec(void* h, bool forward, const float2* in, float2* out, long long rows,
                                           hipStream_t st, int wg_per_cu, int threads, size_t lds_bytes) {
    FusedRegFft* f = static_cast<FusedRegFft*>(h);
    if (rows <= 0) return OFX_OK;
    long long grid = (long long)f->cu_count * wg_per_cu;
    if (grid > rows) grid = rows;
    if (forward) {
        OFX_LDS_ATTR_ONCE(*KFWD, lds_bytes);
        hipLaunchKernelGGL(KFWD, dim3((unsigned)grid), dim3(threads), lds_bytes, st, f->d_t1, f->d_t2, in, out, rows);
    } else {
        OFX_LDS_ATTR_ONCE(*KINV, lds_bytes);
        hipLaunchKernelGGL(KINV, dim3((unsigned)grid), dim3(threads), lds_bytes, st, f->d_t1, f->d_t2, in, out, rows);
    }
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

// -DOFX_STAMPS builds: the stamps of the launch that just ran go to $OFX_STAMP_FILE
// (tools/phase_timeline.py, tools/dev_tail_timeline.py read it).
[[maybe_unused]] static int fused_dump_stamps(hipStream_t st, const void* d_stamps, size_t stamp_bytes) {
    if (const char* f = getenv("OFX_STAMP_FILE")) {
        OFX_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(stamp_bytes / 8);
        OFX_HIP(hipMemcpy(h.data(), d_stamps, stamp_bytes, hipMemcpyDeviceToHost));
        if (FILE* fp = fopen(f, "wb")) {
            fwrite(h.data(), 8, h.size(), fp);
            fclose(fp);
        }
    }
    return OFX_OK;
}
