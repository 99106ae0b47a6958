// ofx_wave.hip -- FUSED engine at 4096 samples (BASELINE configs[0]'s trace length): ONE WAVE PER TRACE.
//
// Path: FeatureExtractors.of1x1_nodelay / _unconstrained / _constrained + baseline / integral /
// maximum / minimum / psd_amp on 4096-sample traces (detprocess/core/algorithms.py:277-570, 650-885,
// 952-1044; processing_data.py:712-772), as ofx_fused.hip does at 32768 samples.
//
// A trace is packed as M = 2048 complex points z[m] = x[2m] + i x[2m+1] and lives in the registers of
// ONE wave: 64 lanes x 32 complex values.  The transform is k_fused's at a smaller geometry,
// M = 16 x 8 x 16 with m = 128 n1 + 16 n2 + n3, k = k1 + 16 k2 + 128 k3:
//
//   load  lane l reads z[128 n1 + l + 64 h], h = 0, 1 (512 contiguous bytes per load of the wave)
//   F1    two 16-point DFTs over n1 per lane, x w_2048^{n' k1}, n' = l + 64 h
//   E1    wave-local LDS transpose  D1[k1][n']
//   F2    four 8-point DFTs over n2 per lane (k1 = (l >> 4) + 4 g, n3 = l & 15), x w_128^{n3 k2}
//   E2    wave-local LDS transpose  D2[k1 + 16 k2][n3]  (row stride 17)
//   F3    lane v owns the 16-point blocks k_low = v and 128 - v (v = 0: the self-paired 0 and 64)
//   mid   slot j pairs bin v + 128 j with M - (v + 128 j): unpack, filter, chi2_0, re-pack in one
//         lane -- the Hermitian-partner layout and lane 0's permutation are those of ofx_fused.hip
//   I3 / E3 / I2 / E4 / I1  mirror images; the full-range fit takes its arg-max from the registers
//         (64 lags per lane; group maxima, then the lane holding the wave's maximum resolves the
//         smallest rolled index), windowed / interpolating searches scan a dump of the 4096 lags in
//         LDS (FEAT bit 0)
//   The next trace is requested into the 64 data registers as soon as the arg-max / dump has read
//   them: the tail (low-frequency chi2, row write) runs under the HBM latency.
//
// What the one-wave geometry buys: NO WORKGROUP BARRIER anywhere -- the four exchanges and the
// arg-max are wave-synchronous (LDS operations of a wave execute in order; __builtin_amdgcn_
// wave_barrier only pins the compiler) -- and 8 independent traces in flight per CU (two 4-wave
// workgroups, 20 KB of LDS per wave) instead of two.  A workgroup's waves share nothing but the
// 1 KB stage-2 twiddle table.  The kernel carries one filter; a plan with several filter slots is one
// launch per slot (ofx_wave_process).
//
// Roofline: HBM, 4096 x 4 + 16 B algorithmic per trace (SURVEY.md section 8d at this length).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ofx_common.h"
#include "ofx_device.h"
#include "ofx_fft_regs.h"
#include "ofx_fused_host.h"

using namespace ofxfft;

namespace {

constexpr int WN = 4096;            // samples
constexpr int WAVES = 4;            // waves (= traces in flight) per workgroup
constexpr int WBLK = 64 * WAVES;
constexpr int WG_PER_CU_W = 2;

#include "ofx_wave_parts.h"
struct WaveShared {
    cpx t2[8 * 16];                 // w_128^{n3 k2}, index k2 * 16 + n3
    WaveLds w[WAVES];
};
static_assert(sizeof(WaveShared) * WG_PER_CU_W <= 160 * 1024, "LDS budget");

struct WaveTabs {
    const float2* t1;     // [6][64]    w_2048^{a n'}, a = 1, 2, 3, 4, 8, 12: the stage-1 anchors
    const float2* t2;     // [8][16]    w_128^{n3 k2}
    const float4* midW;   // [16][64]   (W_k / 2, conj(W_p) / 2)   slot j, lane v
    const float2* midG;   // [16][64]   (g_k', g_p')
    const float2* tbase;  // [64]       T_v = i exp(-2 pi i v / N); T of slot j is T_v w_32^j
    float2 tb0hi;         // base of lane 0 for its slots j >= 8 (block 64)
    float2 wq;            // W_{M/2}
    float gq;             // g_{M/2}
};

[[maybe_unused]] constexpr int NWAVE = 1;        // (stamp layout: one wave per "workgroup")
#include "ofx_fused_stamps.h"
#ifdef OFX_STAMPS                    // diagnostic build: stamps wait for everything outstanding
#define WSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(i); } while (0)
#else
#define WSTAMP(i) STAMP(i)
#endif

// ------------------------------------------------------------------ the kernel
// FEAT bit 0: a windowed / interpolating search (the lags are dumped to LDS); bit 1: time-domain
// windows; bit 2: channel algebra on load.
template <int FEAT>
__global__ __launch_bounds__(WBLK, WG_PER_CU_W) void k_wave(OfxPlanDev pd, OfxSlotDev sd, WaveTabs tabs,
                                                            const float* __restrict__ traces,
                                                            const uint8_t* __restrict__ valid,
                                                            long long n_traces, float* __restrict__ out,
                                                            [[maybe_unused]] unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    WaveShared& SH = *reinterpret_cast<WaveShared*>(smem_raw);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveLds& L = SH.w[wave];
    for (int i = tid; i < 128; i += WBLK) SH.t2[i] = mk(tabs.t2[i].x, tabs.t2[i].y);
    __syncthreads();                    // the only workgroup barrier of the kernel
    const int pre = pd.pre;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(tabs.midW, 16 * 64 * 16);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(tabs.midG, 16 * 64 * 8);
    const size_t ev_stride = (size_t)pd.n_channels * WN;
    cpx* const xc = L.xb;
    float* const xf = reinterpret_cast<float*>(L.xb);
    // roles of this lane (loop-invariant, few)
    const int k1q = lane >> 4, n3 = lane & 15;          // F2: k1 = k1q + 4 g
    const int e1r = k1q * 128 + n3;                     // D1 read base: + 4 g * 128 + 16 n2
    const int e2w = k1q * WLD2 + n3;                    // D2 write base: + (4 g + 16 k2) * WLD2
    const int bB = (lane == 0) ? 64 : 128 - lane;       // partner block of this lane
    // loop-invariant tables of this lane, in registers: stage-1 anchors, T_v, the low-frequency
    // template / weights of its bins
    cpx anch[6];
    {
        const __amdgpu_buffer_rsrc_t rt1 = make_rsrc(tabs.t1, 6 * 64 * 8);
#pragma unroll
        for (int i = 0; i < 6; ++i) anch[i] = buf_ld2(rt1, lane * 8, i * 512);
    }
    cpx tb;
    {
        const __amdgpu_buffer_rsrc_t rtb = make_rsrc(tabs.tbase, 64 * 8);
        tb = buf_ld2(rtb, lane * 8, 0);
    }
    const cpx tbh = (lane == 0) ? mk(tabs.tb0hi.x, tabs.tb0hi.y) : tb;   // (kernel argument: a select)
    constexpr int NLK = WLOW / 64;
    cpx lk_s[NLK];
    float lk_g[NLK];
    {
        const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(sd.s, WLOW * 8);
        const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(sd.g, WLOW * 4);
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            lk_s[i] = mk(0.0f, 0.0f);
            lk_g[i] = 0.0f;
            if (sd.n_search > 0) {
                lk_s[i] = buf_ld2(rs_s, (lane + 64 * i) * 8, 0);
                lk_g[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_g, (lane + 64 * i) * 4, 0, 0));
            }
        }
    }
    bool any_full = false;
    for (int q = 0; q < sd.n_search; ++q) {
        const OfxSearchDev& sq = sd.search[q];
        any_full |= (sq.kind == OFX_SEARCH_DELAY) && !sq.outside && sq.lo == 0 && sq.hi == WN;
    }
    cpx d[WNV];
#ifdef OFX_STAMPS
    int stamp_it = 0;
    unsigned long long* stamp_base = stamps + ((size_t)blockIdx.x * WAVES + wave) * OFX_STAMP_TRACES * 16;
#endif
    // The trace of the NEXT iteration is requested as soon as the registers of this one are free (the
    // tail runs under the loads): the first channel term of the event, raw.
    auto request = [&](long long bb) {
        const float* e = traces + (size_t)bb * ev_stride;
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(e + ((FEAT & 4) ? (size_t)pd.chan[0] * WN : 0), WN * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) d[16 * h + n1] = buf_ld2(rz, (lane + 64 * h) * 8, n1 * 1024);
    };
    const long long wstride = (long long)gridDim.x * WAVES;
    long long b = (long long)blockIdx.x * WAVES + wave;
    bool have = b < n_traces;
    unsigned vcur = 1;
    if (have) {
        if (valid) vcur = valid[b];
        request(b);
    }
    while (have) {
        const long long bnext = b + wstride;
        const bool have_next = bnext < n_traces;
        unsigned vnext = 1;
        float* row = out + (size_t)b * pd.row;
        const long long bcur = b;
        b = bnext;
        have = have_next;
        if (!vcur) {                                    // wave-uniform
            for (int j = lane; j < pd.row; j += 64) row[j] = OFX_SENTINEL;
            if (have_next) {
                if (valid) vnext = valid[bnext];
                request(bnext);
            }
            vcur = vnext;
            continue;
        }
        WSTAMP(0);
        // ------------------------------------------------ channel algebra
        if constexpr (FEAT & 4) {
            if (!(pd.n_terms == 1 && pd.weight[0] == 1.0f)) {
                const float* e = traces + (size_t)bcur * ev_stride;
                const float w0 = pd.weight[0];
#pragma unroll
                for (int j = 0; j < WNV; ++j) d[j] = d[j] * mk(w0, w0);
                for (int c = 1; c < pd.n_terms; ++c) {
                    const __amdgpu_buffer_rsrc_t rc = make_rsrc(e + (size_t)pd.chan[c] * WN, WN * 4);
                    const float wgt = pd.weight[c];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int n1 = 0; n1 < 16; ++n1) {
                            const cpx s = buf_ld2(rc, (lane + 64 * h) * 8, n1 * 1024);
                            d[16 * h + n1] = pfma(mk(wgt, wgt), s, d[16 * h + n1]);
                        }
                }
            }
        }
        // ------------------------------------------------ time-domain windows
        // sample index of d[16 h + n1].{x,y} is 256 n1 + 2 (lane + 64 h) + {0,1}
        if constexpr (FEAT & 2) {
            for (int w = 0; w < pd.n_tdwin; ++w) {
                const int lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
                float s = 0.0f, sq = 0.0f, mx = -INFINITY, mn = INFINITY;
                cpx s2 = mk(0.0f, 0.0f), sq2 = mk(0.0f, 0.0f);
                const unsigned fullm = pd.tdw[w].full, anym = fullm | pd.tdw[w].edge;
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    if (!((anym >> n1) & 1u)) continue;                   // uniform: outside
                    if ((fullm >> n1) & 1u) {                             // uniform: full row
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const cpx v = d[16 * h + n1];
                            s2 = s2 + v;
                            sq2 = pfma(v, v, sq2);
                            mx = max3f(mx, v.x, v.y);
                            mn = min3f(mn, v.x, v.y);
                        }
                    } else {                                              // edge row
                        const int lo_r = lo - WROWS * n1, hi_r = hi - WROWS * n1;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int c = 2 * (lane + 64 * h);
                            const bool in0 = (c >= lo_r) && (c < hi_r);
                            const bool in1 = (c >= lo_r - 1) && (c < hi_r - 1);
                            const cpx v = d[16 * h + n1];
                            const float y0 = in0 ? v.x : 0.0f, y1 = in1 ? v.y : 0.0f;
                            s = (s + y0) + y1;
                            sq = fmaf(y0, y0, fmaf(y1, y1, sq));
                            mx = max3f(mx, in0 ? v.x : -INFINITY, in1 ? v.y : -INFINITY);
                            mn = min3f(mn, in0 ? v.x : INFINITY, in1 ? v.y : INFINITY);
                        }
                    }
                }
                const float S = ofx_wave_sum(s + (s2.x + s2.y));
                const float SQ = ofx_wave_sum(sq + (sq2.x + sq2.y));
                const float MX = ofx_wave_max(mx);
                const float MN = ofx_wave_min(mn);
                if (lane == 0) {
                    const float* e = traces + (size_t)bcur * ev_stride;
                    float first = 0.f, last = 0.f;
                    if constexpr (FEAT & 4) {
                        for (int c = 0; c < pd.n_terms; ++c) {
                            const float* z = e + (size_t)pd.chan[c] * WN;
                            first = fmaf(pd.weight[c], z[lo], first);
                            last = fmaf(pd.weight[c], z[hi - 1], last);
                        }
                    } else {
                        first = e[lo];
                        last = e[hi - 1];
                    }
                    float* o = row + pd.tdw[w].out_off;
                    o[OFX_TD_BASELINE] = S / (float)(hi - lo);
                    o[OFX_TD_INTEGRAL] = (S - 0.5f * (first + last)) * pd.inv_fs;
                    o[OFX_TD_MAXIMUM] = MX;
                    o[OFX_TD_MINIMUM] = MN;
                    o[OFX_TD_SUM] = S;
                    o[OFX_TD_SUMSQ] = SQ;
                    o[OFX_TD_FIRST] = first;
                    o[OFX_TD_LAST] = last;
                }
            }
        }
        if (sd.n_search == 0) {                         // windows only
            if (have_next) {
                if (valid) vnext = valid[bnext];
                request(bnext);
            }
            vcur = vnext;
            continue;
        }
        WSTAMP(1);
        // ---------------------------------------------------------------- F1
        dft<16, -1, WNV, 0>(d);
        dft<16, -1, WNV, 16>(d);
        t1_opaque(anch);
        t1_step<1, false>(d, anch);
        WSTAMP(2);
        // ---------------------------------------------------------------- E1: D1[k1][n']
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) xc[k1 * 128 + lane + 64 * h] = d[16 * h + k1];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) d[8 * g + n2] = xc[e1r + 512 * g + 16 * n2];
        __builtin_amdgcn_wave_barrier();
        WSTAMP(3);
        // ---------------------------------------------------------------- F2
        dft<8, -1, WNV, 0>(d);
        dft<8, -1, WNV, 8>(d);
        dft<8, -1, WNV, 16>(d);
        dft<8, -1, WNV, 24>(d);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k2 = 1; k2 < 8; ++k2) d[8 * g + k2] = cmul(d[8 * g + k2], SH.t2[k2 * 16 + n3]);
        WSTAMP(4);
        // the filter rows of the first middle slots: requested here, ahead of the exchange and F3
        float4 mtw[WMID];
        cpx mtg[WMID];
        wmid_request_first<0>(mtw, mtg, rw, rg, lane);
        // ---------------------------------------------------------------- E2: D2[k1 + 16 k2][n3]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) xc[e2w + (4 * g + 16 * k2) * WLD2] = d[8 * g + k2];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d[j] = xc[lane * WLD2 + j];
            d[16 + j] = xc[bB * WLD2 + j];
        }
        __builtin_amdgcn_wave_barrier();
        WSTAMP(5);
        // ------------------------------------------- F3, middle, I3 (registers)
        dft<16, -1, WNV, 0>(d);
        dft<16, -1, WNV, 16>(d);
        WSTAMP(6);
        cpx chi2v = mk(0.0f, 0.0f);
        {
            const cpx a8 = d[8];
            if (lane == 0) {                            // lane 0: blocks 0 and 64 to the slot shape
#pragma unroll
                for (int j = 0; j < 32; ++j) L.perm[j] = d[j];
                asm volatile("" ::: "memory");          // real LDS reads: no 24-value renaming under EXEC = lane 0
#pragma unroll
                for (int j = 8; j < 32; ++j) d[j] = L.perm[wperm_in_src(j)];
            }
            wmid<0>(d, rw, rg, lane, L, tb, tbh, mtw, mtg, chi2v);
            if (lane == 0) {
                // self-paired bin k = M/2 (A0[8]):  X = conj(Z), Z' = 2 conj(W) Z
                const cpx zq = cmulc(a8, mk(tabs.wq.x, tabs.wq.y));
                chi2v = pfma(a8 * a8, mk(2.0f * tabs.gq, 2.0f * tabs.gq), chi2v);
#pragma unroll
                for (int j = 0; j < 32; ++j) L.perm[j] = d[j];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 9; j < 32; ++j) d[j] = L.perm[wperm_out_src(j)];
                d[8] = zq + zq;
            }
        }
        WSTAMP(7);
        dft<16, +1, WNV, 0>(d);
        dft<16, +1, WNV, 16>(d);
        const float chi0 = ofx_wave_sum(chi2v.x + chi2v.y);
        // ---------------------------------------------------------------- E3
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            xc[lane * WLD2 + j] = d[j];
            xc[bB * WLD2 + j] = d[16 + j];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) d[8 * g + k2] = xc[e2w + (4 * g + 16 * k2) * WLD2];
        __builtin_amdgcn_wave_barrier();
        WSTAMP(8);
        // ---------------------------------------------------------------- I2
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k2 = 1; k2 < 8; ++k2) d[8 * g + k2] = cmulc(d[8 * g + k2], SH.t2[k2 * 16 + n3]);
        dft<8, +1, WNV, 0>(d);
        dft<8, +1, WNV, 8>(d);
        dft<8, +1, WNV, 16>(d);
        dft<8, +1, WNV, 24>(d);
        // ---------------------------------------------------------------- E4
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) xc[e1r + 512 * g + 16 * n2] = d[8 * g + n2];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) d[16 * h + k1] = xc[k1 * 128 + lane + 64 * h];
        __builtin_amdgcn_wave_barrier();
        WSTAMP(9);
        // ---------------------------------------------------------------- I1
        t1_opaque(anch);
        t1_step<1, true>(d, anch);
        dft<16, +1, WNV, 0>(d);
        dft<16, +1, WNV, 16>(d);
        // d[16 h + n1] = (A(n), A(n + 1)), lag n = 256 n1 + 2 (lane + 64 h)
        WSTAMP(10);
        // ------------------------------------------------------------- tail
        // full-range fit from the registers: maximum of A^2 per group of 8 register pairs, then the
        // lane(s) holding the wave's maximum resolve the smallest rolled index among their lags with
        // A^2 == max (NumPy argmin on the rolled chi2: ties -> first index), one group as a rule
        OfxCand fullbest = ofx_cand_none();
        const float a_lag0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(d[0].x)));
        if (any_full) {
            constexpr int NG = WNV / 8;
            float gm[NG];
            float mloc = 0.0f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float m = 0.0f;
#pragma unroll
                for (int j = 8 * g; j < 8 * g + 8; ++j) {
                    const cpx sq = d[j] * d[j];
                    m = max3f(m, sq.x, sq.y);
                }
                gm[g] = m;
                mloc = fmaxf(mloc, m);
            }
            const float Mstar = ofx_wave_max(mloc);
            if (mloc == Mstar) {
                // (the lane id comes from an opaque copy: the 64 rolled indices are recomputed here
                // instead of being hoisted out of the loop and spilled)
                int lt = lane;
                asm volatile("" : "+v"(lt));
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (__builtin_amdgcn_ballot_w64(gm[g] == Mstar) == 0) continue;
                    const int base = 2 * (lt + 64 * (g / 2)) + pre;
#pragma unroll
                    for (int j = 8 * g; j < 8 * g + 8; ++j) {
                        const int n1 = j & 15;
                        const cpx v = d[j];
                        const int i0 = (base + 256 * n1) & (WN - 1);
                        const int i1 = (base + 256 * n1 + 1) & (WN - 1);
                        if (v.x * v.x == Mstar && i0 < fullbest.idx) {
                            fullbest.idx = i0; fullbest.amp = v.x; fullbest.key = Mstar;
                        }
                        if (v.y * v.y == Mstar && i1 < fullbest.idx) {
                            fullbest.idx = i1; fullbest.amp = v.y; fullbest.key = Mstar;
                        }
                    }
                }
            }
            fullbest = ofx_cand_wave_reduce(fullbest);
        }
        if constexpr (FEAT & 1) {                       // the lag dump, natural order
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) xc[128 * n1 + lane + 64 * h] = d[16 * h + n1];
            __builtin_amdgcn_wave_barrier();
        }
        // the registers are free: the next trace is on its way while the tail runs
        if (have_next) {
            if (valid) vnext = valid[bnext];
            request(bnext);
        }
        // psd_amp bands from the stashed 2 X_k (one band after the other: the wave is the workgroup here)
        if (pd.n_bands > 0) {
            const float cpsd = 0.25f / ((float)WN * pd.fs);
            for (int i = 0; i < pd.n_bands; ++i) {
                const int lo = pd.band[i].k_lo, hi = pd.band[i].k_hi;
                float acc = 0.0f;
                for (int k = lo + lane; k < hi; k += 64) {
                    const cpx x2 = L.xlow[k];
                    acc += sqrtf(2.0f * cpsd * fmaf(x2.x, x2.x, x2.y * x2.y));
                }
                acc = ofx_wave_sum(acc);
                if (lane == 0) row[pd.band[i].out_off] = acc / (float)(hi - lo);
            }
        }
        WSTAMP(11);
#pragma unroll 1
        for (int q = 0; q < sd.n_search; ++q) {
            const OfxSearchDev& sq = sd.search[q];
            const bool full = !sq.outside && sq.lo == 0 && sq.hi == WN;
            OfxCand best = ofx_cand_none();
            if (sq.kind == OFX_SEARCH_NODELAY) {
                best.amp = a_lag0;
                best.idx = pre;
                best.key = best.amp * best.amp;
            } else if (full) {
                best = fullbest;
            } else if constexpr (FEAT & 1) {
                // four independent reads per round (reads past the range stay inside the dump)
                auto scan = [&](int i0, int i1) {
                    for (int i = i0 + lane; i < i1; i += 256) {
                        float a[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) a[u] = xf[(i + 64 * u - pre) & (WN - 1)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (i + 64 * u < i1) ofx_cand_take(best, a[u], i + 64 * u);
                    }
                };
                if (sq.outside) {
                    scan(0, sq.lo);
                    scan(sq.hi, WN);
                } else {
                    scan(sq.lo, sq.hi);
                }
                best = ofx_cand_wave_reduce(best);
            }
            OfxRefined ref;
            ref.amp = best.amp;
            ref.frac = 0.0f;
            ref.chi2 = 0.0f;
            bool refine = false;
            if constexpr (FEAT & 1) {
                refine = sq.interp && best.idx != 0x7fffffff;
                if (refine)
                    ref = ofx_interpolate(xf[(best.idx - 1 - pre) & (WN - 1)], best.amp,
                                          xf[(best.idx + 1 - pre) & (WN - 1)], best.idx, WN, sd.norm, chi0);
            }
            // low-frequency chi2: the lane's bins are 64 apart, the phase runs along a chain
            const int dl = best.idx - pre;
            auto phase_of = [&](int k) {
                const int m = (int)(((unsigned)k * (unsigned)dl) & (unsigned)(WN - 1));
                float sn, cs;
                sincospif(-2.0f * ((float)m + (float)k * ref.frac) / (float)WN, &sn, &cs);
                return mk(cs, sn);
            };
            cpx ph = phase_of(lane);
            const cpx step = phase_of(64);
            float low = 0.0f;
#pragma unroll
            for (int i = 0; i < NLK; ++i) {
                const int k = lane + 64 * i;
                if (k < sq.nlow) {
                    const cpx x2 = L.xlow[k];
                    const float pr = ph.x * lk_s[i].x - ph.y * lk_s[i].y;
                    const float pi = ph.x * lk_s[i].y + ph.y * lk_s[i].x;
                    const float rr = 0.5f * x2.x - ref.amp * pr;
                    const float ri = 0.5f * x2.y - ref.amp * pi;
                    low += ((k == 0) ? 1.0f : 2.0f) * lk_g[i] * (rr * rr + ri * ri);
                }
                ph = cmul(ph, step);
            }
            low = ofx_wave_sum(low);
            if (lane == 0) ofx_write_search(row, sq, sd, pd.inv_fs, pre, chi0, best, low, refine ? &ref : nullptr);
        }
        __builtin_amdgcn_wave_barrier();            // the dump is read: the next trace may use the buffer
        WSTAMP(12);
        vcur = vnext;
    }
}

}  // namespace

bool ofx_wave_supported(int n_samples) { return n_samples == WN; }

static int wave_tables(ofx_plan* p) {
    if (p->d_tw1) return OFX_OK;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<float2> t1(6 * 64), t2(8 * 16 + 64);
    const int anchor_mult[6] = {1, 2, 3, 4, 8, 12};
    for (int i = 0; i < 6; ++i)
        for (int n = 0; n < 64; ++n) {
            const double a = -PI2 * (double)((anchor_mult[i] * n) % WM) / WM;
            t1[i * 64 + n] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k2 = 0; k2 < 8; ++k2)
        for (int n3 = 0; n3 < 16; ++n3) {
            const double a = -PI2 * (double)((k2 * n3) % 128) / 128.0;
            t2[k2 * 16 + n3] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int v = 0; v < 64; ++v) {          // tbase[v] = i exp(-2 pi i v / N), stored after t2
        const double a = -PI2 * (double)v / WN;
        t2[8 * 16 + v] = make_float2((float)-std::sin(a), (float)std::cos(a));
    }
    return fused_upload_tables(p, t1, t2);
}

// Middle-step tables of one slot from the fp64 one-sided filter (the recipe of ofx_fused_prepare_slot
// at this geometry): lane v, slot j -> bin k = v + 128 j (lane 0: 128 j for j < 8, 64 + 128 (j - 8)
// above) and its partner M - k.
int ofx_wave_prepare_slot(ofx_plan* p, int slot, const double* wf) {
    int rc = wave_tables(p);
    if (rc) return rc;
    return fused_build_slot_tables(p, slot, wf, WM, 64, 64, 16, [](int v, int j) {
        return v != 0 ? v + 128 * j : (j < 8 ? 128 * j : 64 + 128 * (j - 8));
    });
}

template <int FEAT>
static int launch_wave(ofx_plan* p, const OfxPlanDev& pd, const OfxSlotDev& sd, const WaveTabs& tabs,
                       const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                       hipStream_t st) {
    OFX_LDS_ATTR_ONCE((k_wave<FEAT>), sizeof(WaveShared));
    long long grid = (long long)p->cu_count * WG_PER_CU_W;
    if (grid * WAVES > n) grid = (n + WAVES - 1) / WAVES;
    unsigned long long* d_stamps = nullptr;
#ifdef OFX_STAMPS
    const size_t stamp_bytes = (size_t)grid * WAVES * OFX_STAMP_TRACES * 16 * sizeof(unsigned long long);
    if (p->d_fused_xwide) (void)hipFree(p->d_fused_xwide);
    p->d_fused_xwide = nullptr;
    OFX_HIP(hipMalloc(&p->d_fused_xwide, stamp_bytes));
    OFX_HIP(hipMemset(p->d_fused_xwide, 0, stamp_bytes));
    d_stamps = reinterpret_cast<unsigned long long*>(p->d_fused_xwide);
#endif
    size_t tix = 0;
    int rc = ofx_time_begin(p, st, &tix);
    if (rc) return rc;
    hipLaunchKernelGGL((k_wave<FEAT>), dim3((unsigned)grid), dim3(WBLK), sizeof(WaveShared), st, pd, sd,
                       tabs, d_traces, d_valid, n, d_out, d_stamps);
    rc = ofx_time_end(p, st, tix);
    if (rc) return rc;
    OFX_HIP(hipGetLastError());
#ifdef OFX_STAMPS
    if (int rcd = fused_dump_stamps(st, d_stamps, stamp_bytes)) return rcd;
#endif
    return OFX_OK;
}

int ofx_wave_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                     hipStream_t st) {
    int rc = wave_tables(p);
    if (rc) return rc;
    OfxPlanDev pd;
    ofx_fill_plan_dev(p, &pd);
    struct G { enum { N = WN, ROWS = WROWS, NROWS = 16 }; };
    fused_classify_windows<G>(pd);
    WaveTabs tabs;
    memset(&tabs, 0, sizeof(tabs));
    tabs.t1 = p->d_tw1;
    tabs.t2 = p->d_tw2;
    tabs.tbase = p->d_tw2 + 8 * 16;
    {
        // lane 0, slots j >= 8: bin 64 + 128 (j - 8) = 128 j + (64 - 1024): i exp(-2 pi i (64 - 1024) / N) = -t_64
        const double a = -6.283185307179586476925286766559 * 64.0 / WN;
        tabs.tb0hi = make_float2((float)-std::cos(a), (float)-std::sin(a));
    }
    tabs.midW = reinterpret_cast<const float4*>(p->d_tw1);      // never read without searches
    tabs.midG = p->d_tw1;
    // Validate first (a refused plan must not have launched anything), then one launch per filter
    // slot with searches: the kernel carries one filter, and at this length a second pass over the
    // traces (L2 / MALL hits for the most part) costs less than parking the spectrum would.  The
    // time-domain windows and the bands ride with the first launch.
    int slots[OFX_MAX_SLOTS], nslots = 0;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s) {
        if (!p->slot[s].set || p->slot[s].searches.empty()) continue;
        OfxSlotDev sd;
        ofx_fill_slot_dev(p, s, &sd);
        for (int q = 0; q < sd.n_search; ++q)
            if (sd.search[q].nlow > WLOW) {
                ofx_set_error("FUSED engine (%d samples): lowchi2_fcutoff covers %d bins (> %d)", WN,
                              sd.search[q].nlow, WLOW);
                return OFX_ERR_UNSUPPORTED;
            }
        slots[nslots++] = s;
    }
    if (pd.n_bands > 0) {
        if (nslots == 0) {
            ofx_set_error("FUSED engine: psd_amp bands need at least one filter slot with a "
                          "search on the plan (use the ROCFFT engine otherwise)");
            return OFX_ERR_UNSUPPORTED;
        }
        for (int i = 0; i < pd.n_bands; ++i)
            if (pd.band[i].k_hi > WLOW) {
                ofx_set_error("FUSED engine (%d samples): band [%d,%d) exceeds the %d stashed bins", WN,
                              pd.band[i].k_lo, pd.band[i].k_hi, WLOW);
                return OFX_ERR_UNSUPPORTED;
            }
    }
    for (int li = 0; li < (nslots > 0 ? nslots : 1); ++li) {
        OfxSlotDev sd;
        memset(&sd, 0, sizeof(sd));
        if (nslots > 0) {
            const int s = slots[li];
            ofx_fill_slot_dev(p, s, &sd);
            tabs.midW = p->slot[s].d_pq;
            tabs.midG = reinterpret_cast<const float2*>(p->slot[s].d_pq + 16 * 64);
            tabs.wq = make_float2(p->slot[s].wq_x, p->slot[s].wq_y);
            tabs.gq = p->slot[s].gq;
        }
        if (li == 1) pd.n_tdwin = pd.n_bands = 0;
        int feat = 0;
        for (int q = 0; q < sd.n_search; ++q) {
            const OfxSearchDev& sq = sd.search[q];
            const bool full = sq.lo == 0 && sq.hi == WN && !sq.outside;
            if (sq.kind == OFX_SEARCH_DELAY && (sq.interp || !full)) feat |= 1;
        }
        if (pd.n_tdwin > 0) feat |= 2;
        if (p->n_channels > 1 || p->n_terms > 1 || p->weight[0] != 1.0) feat |= 4;
        switch (feat) {
#define OFX_CASE(F) case F: rc = launch_wave<F>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st); break;
            OFX_CASE(0) OFX_CASE(1) OFX_CASE(2) OFX_CASE(3) OFX_CASE(4) OFX_CASE(5) OFX_CASE(6)
#undef OFX_CASE
            default: rc = launch_wave<7>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st);
        }
        if (rc) return rc;
    }
    return OFX_OK;
}
