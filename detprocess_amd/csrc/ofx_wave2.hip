// ofx_wave2.hip -- FUSED engine at 8192 and 16384 samples: TWO / FOUR WAVES PER TRACE.
//
// Path: FeatureExtractors.of1x1_nodelay / _unconstrained / _constrained + baseline / integral /
// maximum / minimum / psd_amp on 8192- and 16384-sample traces (detprocess/core/algorithms.py:277-570, 650-885,
// 952-1044; processing_data.py:712-772), as ofx_wave.hip does at 4096 samples.
//
// W waves per trace (W = 2: 8192 samples, W = 4: 16384).  The packed transform has M' = 2048 W complex points;
// it is split once, decimation in frequency, into W transforms of 2048 points -- exactly the transform a wave of
// ofx_wave.hip runs in its registers:
//
//   y_s[m] = (sum_j z[m + 2048 j] e^{-2 pi i s j / W}) w_M'^{m s},  s = 0 .. W-1      X[W q + s] = FFT_2048(y_s)[q]
//
// Wave s loads its own part z[m + 2048 s] (and sums the time-domain windows over it); the parts meet through
// the waves' exchange buffers in LDS, every wave combining them at its own (lane, register), and from there
// on wave s owns the bins of residue s: F1 / E1 / F2 / E2 / F3 of ofx_wave_parts.h, no barrier, no exchange.
// The Hermitian partner of bin W q + s is M' - (W q + s) = W (2048 - q) [s = 0] or W (2047 - q) + (W - s):
//   s = 0      pairs inside the wave: the layout of k_wave, DC / Nyquist and the self-paired bin M'/2 in lane 0
//   s = W / 2  pairs inside the wave: blocks v and 127 - v, nothing self-paired
//   s = 1, 3 (W = 4)  cross: lane v of wave 1 holds the blocks v and 127 - v of its bins and needs block 127 - v
//              of wave 3 -- the partner half of wave 3's lane v.  The two waves swap the partner halves of their
//              lanes through LDS before the middle step and swap the results back after it (lane to lane, row
//              stride 17, three workgroup barriers); their partners' low bins go to the other wave's stash.
// The inverse mirrors the forward; then the waves meet again,
//
//   A(2 m + e + 4096 j) = Re / Im of  sum_s conj(w_M'^{m s}) y'_s[m] e^{+2 pi i s j / W}
//
// and wave j holds the lags of part j.  Arg-max candidates, chi2_0, the window sums, the band sums and the
// low-frequency chi2 (wave s owns the bins of residue s) are partial per wave and meet in a few words of LDS,
// double-buffered by trace parity; windowed searches scan the dump of all parts; wave 0 writes the row.  One
// trace per workgroup, eight waves per CU.  The next trace's part is requested into 64 registers as soon as the
// lags have been reduced / dumped.
//
// Roofline: HBM, N x 4 + 16 B algorithmic per trace.
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

// W waves per trace: N = 4096 W samples, W x 2048 packed complex points, one trace per workgroup of W waves,
// eight waves per CU

#ifndef OFX_WMID_DEPTH
#define OFX_WMID_DEPTH 2            // (8 in k_wave: here the raw parts of the next trace need the registers; measured:
                                    //  2 / 3 / 4 slots ahead = 22.5 / 22.1 / 21.8 M traces/s at 16384 samples, no
                                    //  difference at 8192)
#endif
#include "ofx_wave_parts.h"

template <int W>
struct Wave2X {                     // what the waves of a trace tell each other (double-buffered by parity)
    float chi[W];
    float lag0;
    OfxCand cand[W];
    float td[OFX_MAX_TDWIN][4][W];
    float lowp[OFX_MAX_SEARCHES][W];
    float band[OFX_MAX_BANDS][W];
};
template <int W>
struct Wave2Shared {
    WaveLds w[W];
    Wave2X<W> x[2];
};
static_assert(sizeof(Wave2Shared<2>) * 4 <= 160 * 1024, "LDS budget");
static_assert(sizeof(Wave2Shared<4>) * 2 <= 160 * 1024, "LDS budget");
constexpr int LDS_WAVE_FLOATS = sizeof(WaveLds) / 4;      // from w[0].xb to w[1].xb, in floats

struct Wave2Tabs {
    const float2* t1;     // [6][64]     w_2048^{a n'}, a = 1, 2, 3, 4, 8, 12: the stage-1 anchors
    const float2* t2;     // [8][16]     w_128^{n3 k2}: seven per lane, kept in registers
    const float2* ua;     // [W][2][64]  w_M'^{(n + 64 h) s}: the split twiddle of wave s (x a 64th root per row n1)
    const float2* tbase;  // [W][64]     T(v, s) = i exp(-2 pi i (W v + s) / N); slot j: T w_32^j
    const float4* midW;   // [W][16][64] (W_k / 2, conj(W_p) / 2), k = W (v + 128 j) + s
    const float2* midG;   // [W][16][64] (g_k', g_p')
    float2 tb0hi;         // wave 0, lane 0, slots j >= 8 (block 64)
    float2 wq;            // W_{M'/2}
    float gq;             // g_{M'/2}
};

// cos(2 pi j / 64), j = 0 .. 16
constexpr double kCos64[17] = {
    1.00000000000000000000, 0.99518472667219692873, 0.98078528040323043058, 0.95694033573220882438,
    0.92387953251128673848, 0.88192126434835504956, 0.83146961230254523567, 0.77301045336273699338,
    0.70710678118654757274, 0.63439328416364548779, 0.55557023301960228867, 0.47139673682599780857,
    0.38268343236508983729, 0.29028467725446233105, 0.19509032201612833135, 0.09801714032956077016,
    0.0};
__host__ __device__ constexpr double cos64(int j) {
    j = ((j % 64) + 64) % 64;
    if (j <= 16) return kCos64[j];
    if (j <= 32) return -kCos64[32 - j];
    if (j <= 48) return -kCos64[j - 32];
    return kCos64[64 - j];
}
__host__ __device__ constexpr double sin64(int j) { return cos64(j - 16); }
// b * exp(DIR * 2 pi i NUM / 64), NUM a compile-time constant
template <int NUM, int DIR>
__device__ __forceinline__ cpx rot64(cpx b) {
    constexpr int n = ((NUM % 64) + 64) % 64;
    if constexpr (n % 2 == 0) {
        return twmul<n / 2, DIR>(b);
    } else {
        constexpr float c = (float)cos64(n);
        constexpr float s = (float)(DIR * sin64(n));
        return pfma(swp(b), mk(-s, s), b * mk(c, c));
    }
}

// the split twiddle of wave s, forward (INV = false: x w_M'^{m s}) and back (x conj): per element the lane's
// anchor u_h = w_M'^{(lane + 64 h) s} and the row's constant w_M'^{128 n1 s} = exp(-2 pi i n1 S4 / 64), S4 = 4 s / W
template <int S4, int N1, bool INV>
__device__ __forceinline__ void split_tw(cpx (&d)[WNV], cpx u0, cpx u1) {
    if constexpr (N1 < 16) {
        if constexpr (!INV) {
            d[N1] = rot64<N1 * S4, -1>(cmul(d[N1], u0));
            d[16 + N1] = rot64<N1 * S4, -1>(cmul(d[16 + N1], u1));
        } else {
            d[N1] = rot64<N1 * S4, +1>(cmulc(d[N1], u0));
            d[16 + N1] = rot64<N1 * S4, +1>(cmulc(d[16 + N1], u1));
        }
        split_tw<S4, N1 + 1, INV>(d, u0, u1);
    }
}
template <int W, bool INV>
__device__ __forceinline__ void split_tw_of(int s, cpx (&d)[WNV], cpx u0, cpx u1) {
    if (s == 1) split_tw<4 / W, 0, INV>(d, u0, u1);
    if constexpr (W == 4) {
        if (s == 2) split_tw<2, 0, INV>(d, u0, u1);
        if (s == 3) split_tw<3, 0, INV>(d, u0, u1);
    }
}

// ------------------------------------------------------------------ the kernel
// FEAT bit 0: a windowed / interpolating search (the lags are dumped to LDS); bit 1: time-domain
// windows; bit 2: channel algebra on load.
template <int W, int FEAT>
__global__ __launch_bounds__(64 * W, 2) void k_wave2(OfxPlanDev pd, OfxSlotDev sd, Wave2Tabs tabs,
                                                     const float* __restrict__ traces,
                                                     const uint8_t* __restrict__ valid, long long n_traces,
                                                     float* __restrict__ out) {
    constexpr int VN = 4096 * W;        // samples
    constexpr int VBLK = 64 * W;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Wave2Shared<W>& SH = *reinterpret_cast<Wave2Shared<W>*>(smem_raw);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int s = __builtin_amdgcn_readfirstlane(tid >> 6);       // residue of this wave's bins mod W / its part of the lags
    WaveLds& L = SH.w[s];
    WaveLds& LO = SH.w[(W - s) % W == s ? s ^ 1 : (W - s) % W];   // W = 2: the other wave; W = 4: the partner of a crossing wave
    const bool crossing = (W == 4) && (s & 1);                    // bins W q + s pair with W (2047 - q) + (W - s)
    const int pre = pd.pre;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(tabs.midW + s * 16 * 64, 16 * 64 * 16);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(tabs.midG + s * 16 * 64, 16 * 64 * 8);
    const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(sd.s, W * WLOW * 8);
    const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(sd.g, W * WLOW * 4);
    const size_t ev_stride = (size_t)pd.n_channels * VN;
    cpx* const xc = L.xb;
    const float* const xf0 = reinterpret_cast<const float*>(SH.w[0].xb);      // the dump: lag n at
                                                                              // xf0[(n >> 12) LDS_WAVE_FLOATS + (n & 4095)]
    // roles of this lane (loop-invariant, few)
    const int k1q = lane >> 4, n3 = lane & 15;          // F2: k1 = k1q + 4 g
    const int e1r = k1q * 128 + n3;                     // D1 read base: + 4 g * 128 + 16 n2
    const int e2w = k1q * WLD2 + n3;                    // D2 write base: + (4 g + 16 k2) * WLD2
    const int bB = s ? 127 - lane : (lane == 0 ? 64 : 128 - lane);     // partner block of this lane
    // loop-invariant tables of this lane, in registers
    cpx anch[6], t2r[7];
    {
        const __amdgpu_buffer_rsrc_t rt1 = make_rsrc(tabs.t1, 6 * 64 * 8);
#pragma unroll
        for (int i = 0; i < 6; ++i) anch[i] = buf_ld2(rt1, lane * 8, i * 512);
        const __amdgpu_buffer_rsrc_t rt2 = make_rsrc(tabs.t2, 128 * 8);
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2) t2r[k2 - 1] = buf_ld2(rt2, n3 * 8, k2 * 128);
    }
    cpx tb, u0, u1;
    {
        const __amdgpu_buffer_rsrc_t rtb = make_rsrc(tabs.tbase, W * 64 * 8);
        tb = buf_ld2(rtb, lane * 8, s * 512);
        const __amdgpu_buffer_rsrc_t rua = make_rsrc(tabs.ua, W * 128 * 8);
        u0 = buf_ld2(rua, lane * 8, s * 1024);
        u1 = buf_ld2(rua, lane * 8, s * 1024 + 512);
    }
    const cpx tbh = (s == 0 && lane == 0) ? mk(tabs.tb0hi.x, tabs.tb0hi.y) : tb;
    bool any_full = false;
    for (int q = 0; q < sd.n_search; ++q) {
        const OfxSearchDev& sq = sd.search[q];
        any_full |= (sq.kind == OFX_SEARCH_DELAY) && !sq.outside && sq.lo == 0 && sq.hi == VN;
    }
    // the wave's own part of the trace, a = z[m + 2048 s], m = 128 n1 + lane + 64 h: the parts meet through LDS
    // (every wave reading the whole trace was tried first: four readers per line were four HBM reads at W = 4,
    // 328 KB of traffic per 64 KB trace; at W = 2 the second read hit L2 and the two forms run at the same rate)
    cpx a[WNV];
    cpx d[WNV];
    auto load_part = [&](cpx (&z)[WNV], const float* chan, int part) {
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(chan, VN * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) z[16 * h + n1] = buf_ld2(rz, (lane + 64 * h) * 8, n1 * 1024 + part * 16384);
    };
    auto request = [&](long long bb) {
        load_part(a, traces + (size_t)bb * ev_stride + ((FEAT & 4) ? (size_t)pd.chan[0] * VN : 0), s);
    };
    // channel algebra on one part (FEAT bit 2): weight of the first term, then the other terms
    [[maybe_unused]] auto combine_part = [&](cpx (&z)[WNV], long long bb, int part) {
        if (pd.n_terms == 1 && pd.weight[0] == 1.0f) return;
        const float* e = traces + (size_t)bb * ev_stride;
        const float w0 = pd.weight[0];
#pragma unroll
        for (int j = 0; j < WNV; ++j) z[j] = z[j] * mk(w0, w0);
        for (int c = 1; c < pd.n_terms; ++c) {
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(e + (size_t)pd.chan[c] * VN, VN * 4);
            const float wgt = pd.weight[c];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const cpx zz = buf_ld2(rc, (lane + 64 * h) * 8, n1 * 1024 + part * 16384);
                    z[16 * h + n1] = pfma(mk(wgt, wgt), zz, z[16 * h + n1]);
                }
        }
    };
    long long b = (long long)blockIdx.x;
    bool have = b < n_traces;
    unsigned vcur = 1;
    int par = 0;
    if (have) {
        if (valid) vcur = valid[b];
        request(b);
    }
    while (have) {
        const long long bnext = b + (long long)gridDim.x;
        const bool have_next = bnext < n_traces;
        unsigned vnext = 1;
        float* row = out + (size_t)b * pd.row;
        const long long bcur = b;
        b = bnext;
        have = have_next;
        if (!vcur) {                                    // uniform over the workgroup
            for (int j = tid; j < pd.row; j += VBLK) row[j] = OFX_SENTINEL;
            if (have_next) {
                if (valid) vnext = valid[bnext];
                request(bnext);
            }
            vcur = vnext;
            continue;
        }
        par ^= 1;
        Wave2X<W>& X = SH.x[par];
        // ------------------------------------------------ channel algebra
        if constexpr (FEAT & 4) {
            combine_part(a, bcur, s);
        }
        // ------------------------------------------------ time-domain windows
        // wave s sums over its part of the samples: index 4096 s + 256 n1 + 2 (lane + 64 h) + {0, 1}, rows
        // 16 s + n1 of the host's classification; the partial sums meet in X.td
        [[maybe_unused]] auto td_sums = [&](const cpx (&z)[WNV]) {
            for (int w = 0; w < pd.n_tdwin; ++w) {
                const int lo = pd.tdw[w].lo - 4096 * s, hi = pd.tdw[w].hi - 4096 * s;
                float sm = 0.0f, sq = 0.0f, mx = -INFINITY, mn = INFINITY;
                cpx s2 = mk(0.0f, 0.0f), sq2 = mk(0.0f, 0.0f);
                // rows of this part: 32 rows classified by the host (W = 2); 64 rows, classified here (W = 4)
                unsigned fullm, anym;
                if constexpr (W == 2) {
                    fullm = pd.tdw[w].full >> (16 * s);
                    anym = fullm | (pd.tdw[w].edge >> (16 * s));
                } else {
                    fullm = anym = 0;
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const int r0 = WROWS * n1;
                        if (r0 < hi && r0 + WROWS > lo) anym |= 1u << n1;
                        if (lo <= r0 && r0 + WROWS <= hi) fullm |= 1u << n1;
                    }
                }
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    if (!((anym >> n1) & 1u)) continue;                   // uniform: outside
                    if ((fullm >> n1) & 1u) {                             // uniform: full row
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const cpx v = z[16 * h + n1];
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
                            const cpx v = z[16 * h + n1];
                            const float y0 = in0 ? v.x : 0.0f, y1 = in1 ? v.y : 0.0f;
                            sm = (sm + y0) + y1;
                            sq = fmaf(y0, y0, fmaf(y1, y1, sq));
                            mx = max3f(mx, in0 ? v.x : -INFINITY, in1 ? v.y : -INFINITY);
                            mn = min3f(mn, in0 ? v.x : INFINITY, in1 ? v.y : INFINITY);
                        }
                    }
                }
                const float S = ofx_wave_sum(sm + (s2.x + s2.y));
                const float SQ = ofx_wave_sum(sq + (sq2.x + sq2.y));
                const float MX = ofx_wave_max(mx);
                const float MN = ofx_wave_min(mn);
                if (lane == 0) {
                    X.td[w][0][s] = S;
                    X.td[w][1][s] = MX;
                    X.td[w][2][s] = MN;
                    X.td[w][3][s] = SQ;
                }
            }
        };
        if constexpr (FEAT & 2) td_sums(a);
        // one lane of wave 0 per window: the partials, the end points, the eight values
        auto td_finalize = [&]() {
            if constexpr (FEAT & 2) {
                if (s == 0 && lane < pd.n_tdwin) {
                    const int w = lane;
                    const int lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
                    const float* e = traces + (size_t)bcur * ev_stride;
                    float first = 0.f, last = 0.f;
                    if constexpr (FEAT & 4) {
                        for (int c = 0; c < pd.n_terms; ++c) {
                            const float* z = e + (size_t)pd.chan[c] * VN;
                            first = fmaf(pd.weight[c], z[lo], first);
                            last = fmaf(pd.weight[c], z[hi - 1], last);
                        }
                    } else {
                        first = e[lo];
                        last = e[hi - 1];
                    }
                    float S = 0.0f, SQ = 0.0f, MX = -INFINITY, MN = INFINITY;
#pragma unroll
                    for (int t = 0; t < W; ++t) {
                        S += X.td[w][0][t];
                        MX = fmaxf(MX, X.td[w][1][t]);
                        MN = fminf(MN, X.td[w][2][t]);
                        SQ += X.td[w][3][t];
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
        };
        // ---------------------------------------------------------------- the split: y_s
        {
            if (sd.n_search == 0) {                     // windows only
                __syncthreads();
                td_finalize();
                if (have_next) {
                    if (valid) vnext = valid[bnext];
                    request(bnext);
                }
                vcur = vnext;
                continue;
            }
            // the parts meet: y_s before its twiddle = sum_j z_j (-i)^{s j} (W = 2: z_0 +- z_1), every wave at its own
            // (lane, register)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) xc[128 * n1 + lane + 64 * h] = a[16 * h + n1];
            __syncthreads();
            if constexpr (W == 2) {
                const cpx* const x0 = SH.w[0].xb;
                const cpx* const x1 = SH.w[1].xb;
                const float sg = s ? -1.0f : 1.0f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const int m = 128 * n1 + lane + 64 * h;
                        d[16 * h + n1] = pfma(x1[m], mk(sg, sg), x0[m]);
                    }
            } else {
                const cpx* const x0 = SH.w[0].xb;
                const cpx* const x1 = SH.w[1].xb;
                const cpx* const x2 = SH.w[W - 2].xb;
                const cpx* const x3 = SH.w[W - 1].xb;
                const float sg = (s & 1) ? -1.0f : 1.0f;    // (-i)^{2 s}
                const float sr = (s & 2) ? -1.0f : 1.0f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const int m = 128 * n1 + lane + 64 * h;
                        const cpx e02 = pfma(x2[m], mk(sg, sg), x0[m]);
                        const cpx e13 = pfma(x3[m], mk(sg, sg), x1[m]);
                        // (-i)^s = 1, -i, -1, i:  s odd -> (y, -x) sr ; s even -> sr
                        d[16 * h + n1] = (s & 1) ? pfma(swp(e13), mk(sr, -sr), e02) : pfma(e13, mk(sr, sr), e02);
                        // (four reads per element: the fence ties the four results of a row group to their
                        // place -- without it the compiler reads all 128 values first and combines them after
                        // the barrier: 256 registers of LDS data, 150 of them spilled, in the first build)
                        if ((n1 & 3) == 3)
                            asm volatile("" : "+v"(d[16 * h + n1 - 3]), "+v"(d[16 * h + n1 - 2]), "+v"(d[16 * h + n1 - 1]),
                                         "+v"(d[16 * h + n1]) :: "memory");
                    }
            }
            __syncthreads();                            // the buffers are the waves' own again
        }
        if (s != 0) {
            cpx uu0 = u0, uu1 = u1;
            asm volatile("" : "+v"(uu0), "+v"(uu1));    // (no hoisting of the 32 element twiddles out of the loop)
            split_tw_of<W, false>(s, d, uu0, uu1);
        }
        // ---------------------------------------------------------------- F1
        dft<16, -1, WNV, 0>(d);
        dft<16, -1, WNV, 16>(d);
        t1_opaque(anch);
        t1_step<1, false>(d, anch);
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
        // ---------------------------------------------------------------- F2
        dft<8, -1, WNV, 0>(d);
        dft<8, -1, WNV, 8>(d);
        dft<8, -1, WNV, 16>(d);
        dft<8, -1, WNV, 24>(d);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k2 = 1; k2 < 8; ++k2) d[8 * g + k2] = cmul(d[8 * g + k2], t2r[k2 - 1]);
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
        // ------------------------------------------- F3, middle, I3 (registers)
        dft<16, -1, WNV, 0>(d);
        dft<16, -1, WNV, 16>(d);
        // W = 4, waves 1 and 3: the partner halves of the lanes change hands (bins 4 q + 1 pair with
        // 4 (2047 - q) + 3): lane to lane, row stride 17, first half of the exchange buffer
        if constexpr (W == 4) {
            if (crossing) {
#pragma unroll
                for (int j = 0; j < 16; ++j) xc[lane * WLD2 + j] = d[16 + j];
            }
            __syncthreads();
            if (crossing) {
#pragma unroll
                for (int j = 0; j < 16; ++j) d[16 + j] = LO.xb[lane * WLD2 + j];
            }
        }
        cpx chi2v = mk(0.0f, 0.0f);
        {
            int lm = lane;
            cpx tlo = tb, thi = tbh;
            asm volatile("" : "+v"(lm), "+v"(tlo), "+v"(thi));
            if (s == 0) {
                const cpx a8 = d[8];
                if (lane == 0) {                        // lane 0: blocks 0 and 64 to the slot shape
#pragma unroll
                    for (int j = 0; j < 32; ++j) L.perm[j] = d[j];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 8; j < 32; ++j) d[j] = L.perm[wperm_in_src(j)];
                }
                wmid<0, 0>(d, rw, rg, lm, L, tlo, thi, mtw, mtg, chi2v);
                if (lane == 0) {
                    // self-paired bin k = M'/2 (A0[8]):  X = conj(Z), Z' = 2 conj(W) Z
                    const cpx zq = cmulc(a8, mk(tabs.wq.x, tabs.wq.y));
                    chi2v = pfma(a8 * a8, mk(2.0f * tabs.gq, 2.0f * tabs.gq), chi2v);
#pragma unroll
                    for (int j = 0; j < 32; ++j) L.perm[j] = d[j];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 9; j < 32; ++j) d[j] = L.perm[wperm_out_src(j)];
                    d[8] = zq + zq;
                }
            } else if (!crossing) {
                wmid<0, 1>(d, rw, rg, lm, L, tlo, thi, mtw, mtg, chi2v);
            } else {
                wmid<0, 2>(d, rw, rg, lm, L, tlo, thi, mtw, mtg, chi2v, &LO);
            }
        }
        if constexpr (W == 4) {                         // ... and back (second half of the buffer)
            if (crossing) {
#pragma unroll
                for (int j = 0; j < 16; ++j) xc[(64 + lane) * WLD2 + j] = d[16 + j];
            }
            __syncthreads();
            if (crossing) {
#pragma unroll
                for (int j = 0; j < 16; ++j) d[16 + j] = LO.xb[(64 + lane) * WLD2 + j];
            }
            __syncthreads();                            // the buffers are the waves' own again
        }
        dft<16, +1, WNV, 0>(d);
        dft<16, +1, WNV, 16>(d);
        const float chi_w = ofx_wave_sum(chi2v.x + chi2v.y);
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
        // ---------------------------------------------------------------- I2
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k2 = 1; k2 < 8; ++k2) d[8 * g + k2] = cmulc(d[8 * g + k2], t2r[k2 - 1]);
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
        // ---------------------------------------------------------------- I1
        t1_opaque(anch);
        t1_step<1, true>(d, anch);
        dft<16, +1, WNV, 0>(d);
        dft<16, +1, WNV, 16>(d);
        // d[16 h + n1] = y'_s[m], m = 128 n1 + lane + 64 h
        // ---------------------------------------------------------------- the parts meet
        if (s != 0) {
            cpx uu0 = u0, uu1 = u1;
            asm volatile("" : "+v"(uu0), "+v"(uu1));
            split_tw_of<W, true>(s, d, uu0, uu1);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) xc[128 * n1 + lane + 64 * h] = d[16 * h + n1];
        if (lane == 0) X.chi[s] = chi_w;
        __syncthreads();                                // B1: every u_t = conj(w^{m t}) y'_t is in LDS
        if constexpr (W == 2) {
            const cpx* const xo = LO.xb;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const cpx o = xo[128 * n1 + lane + 64 * h];
                    d[16 * h + n1] = s ? o - d[16 * h + n1] : d[16 * h + n1] + o;
                }
        } else {
            // part s of the lags: sum_t u_t i^{t s}
            const cpx* const x0 = SH.w[0].xb;
            const cpx* const x1 = SH.w[1].xb;
            const cpx* const x2 = SH.w[2].xb;
            const cpx* const x3 = SH.w[3].xb;
            const float sg = (s & 1) ? -1.0f : 1.0f;    // i^{2 s}
            const float sr = (s & 2) ? -1.0f : 1.0f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const int m = 128 * n1 + lane + 64 * h;
                    const cpx e02 = pfma(x2[m], mk(sg, sg), x0[m]);        // u0 + i^{2s} u2
                    const cpx e13 = pfma(x3[m], mk(sg, sg), x1[m]);        // u1 + i^{2s} u3 (times i^s below)
                    // i^s = 1, i, -1, -i:  s odd -> (-y, x) sr ; s even -> sr
                    d[16 * h + n1] = (s & 1) ? pfma(swp(e13), mk(-sr, sr), e02) : pfma(e13, mk(sr, sr), e02);
                    if ((n1 & 3) == 3)
                        asm volatile("" : "+v"(d[16 * h + n1 - 3]), "+v"(d[16 * h + n1 - 2]), "+v"(d[16 * h + n1 - 1]),
                                     "+v"(d[16 * h + n1]) :: "memory");
                }
        }
        // d[16 h + n1] = (A(n), A(n + 1)), lag n = 4096 s + 256 n1 + 2 (lane + 64 h)
        __syncthreads();                                // B2: the exchange buffers are free again
        // ------------------------------------------------------------- tail
        int lt = lane;
        asm volatile("" : "+v"(lt));
        constexpr int NLK = WLOW / 64;
        cpx lk_s[NLK];
        float lk_g[NLK];
#pragma unroll
        for (int i = 0; i < NLK; ++i) {                 // this wave's low bins: k = W (lane + 64 i) + s
            lk_s[i] = buf_ld2(rs_s, (W * (lt + 64 * i) + s) * 8, 0);
            lk_g[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_g, (W * (lt + 64 * i) + s) * 4, 0, 0));
        }
        OfxCand mybest = ofx_cand_none();
        if (s == 0 && lane == 0) X.lag0 = d[0].x;
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
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (__builtin_amdgcn_ballot_w64(gm[g] == Mstar) == 0) continue;
                    const int base = 2 * (lt + 64 * (g / 2)) + pre + 4096 * s;
#pragma unroll
                    for (int j = 8 * g; j < 8 * g + 8; ++j) {
                        const int n1 = j & 15;
                        const cpx v = d[j];
                        const int i0 = (base + 256 * n1) & (VN - 1);
                        const int i1 = (base + 256 * n1 + 1) & (VN - 1);
                        if (v.x * v.x == Mstar && i0 < mybest.idx) {
                            mybest.idx = i0; mybest.amp = v.x; mybest.key = Mstar;
                        }
                        if (v.y * v.y == Mstar && i1 < mybest.idx) {
                            mybest.idx = i1; mybest.amp = v.y; mybest.key = Mstar;
                        }
                    }
                }
            }
            mybest = ofx_cand_wave_reduce(mybest);
        }
        if (lane == 0) X.cand[s] = mybest;
        if constexpr (FEAT & 1) {                       // the lag dump: this wave's half, natural order
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) xc[128 * n1 + lt + 64 * h] = d[16 * h + n1];
        }
        // the registers are free: the next trace is on its way while the tail runs (requested without a
        // condition -- the last trace of a workgroup reads itself again --: a conditional request would
        // keep the 128 registers of the two halves alive through the whole body)
        {
            const long long bq_ = have_next ? bnext : bcur;
            if (valid) vnext = valid[bq_];
            request(bq_);
        }
        // psd_amp bands: this wave's bins (k of its parity), partial sums
        if (pd.n_bands > 0) {
            const float cpsd = 0.25f / ((float)VN * pd.fs);
            for (int i = 0; i < pd.n_bands; ++i) {
                const int lo = pd.band[i].k_lo, hi = pd.band[i].k_hi;
                float acc = 0.0f;
                for (int k = lo + lt; k < hi; k += 64) {
                    if ((k & (W - 1)) == s) {
                        const cpx x2 = L.xlow[k / W];
                        acc += sqrtf(2.0f * cpsd * fmaf(x2.x, x2.x, x2.y * x2.y));
                    }
                }
                acc = ofx_wave_sum(acc);
                if (lane == 0) X.band[i][s] = acc;
            }
        }
        __syncthreads();                                // B3: chi2_0, candidates, lag 0, window and band partials, dump
        float chi0 = X.chi[0];
        OfxCand fullbest = X.cand[0];
#pragma unroll
        for (int t = 1; t < W; ++t) {
            chi0 += X.chi[t];
            const OfxCand c1 = X.cand[t];
            if (ofx_cand_better(c1.key, c1.idx, fullbest)) fullbest = c1;
        }
        const float a_lag0 = X.lag0;
        td_finalize();
        if (s == 0 && lane == 0)
            for (int i = 0; i < pd.n_bands; ++i) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < W; ++t) acc += X.band[i][t];
                row[pd.band[i].out_off] = acc / (float)(pd.band[i].k_hi - pd.band[i].k_lo);
            }
#pragma unroll 1
        for (int q = 0; q < sd.n_search; ++q) {
            const OfxSearchDev& sq = sd.search[q];
            const bool full = !sq.outside && sq.lo == 0 && sq.hi == VN;
            OfxCand best = ofx_cand_none();
            if (sq.kind == OFX_SEARCH_NODELAY) {
                best.amp = a_lag0;
                best.idx = pre;
                best.key = best.amp * best.amp;
            } else if (full) {
                best = fullbest;
            } else if constexpr (FEAT & 1) {
                // both waves scan the whole range (the dump of either half is readable by both): the same
                // winner in both, nothing to combine
                auto lagv = [&](int i) {
                    const int n = (i - pre) & (VN - 1);
                    return xf0[(n >> 12) * LDS_WAVE_FLOATS + (n & 4095)];
                };
                auto scan = [&](int i0, int i1) {
                    for (int i = i0 + lt; i < i1; i += 256) {
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = lagv(i + 64 * u);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (i + 64 * u < i1) ofx_cand_take(best, v[u], i + 64 * u);
                    }
                };
                if (sq.outside) {
                    scan(0, sq.lo);
                    scan(sq.hi, VN);
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
                if (refine) {
                    const int nm = (best.idx - 1 - pre) & (VN - 1), np = (best.idx + 1 - pre) & (VN - 1);
                    ref = ofx_interpolate(xf0[(nm >> 12) * LDS_WAVE_FLOATS + (nm & 4095)], best.amp,
                                          xf0[(np >> 12) * LDS_WAVE_FLOATS + (np & 4095)], best.idx, VN, sd.norm,
                                          chi0);
                }
            }
            // low-frequency chi2: this wave's bins k = W (lane + 64 i) + s, the phase along a chain
            const int dl = best.idx - pre;
            auto phase_of = [&](int k) {
                const int m = (int)(((unsigned)k * (unsigned)dl) & (unsigned)(VN - 1));
                float sn, cs;
                sincospif(-2.0f * ((float)m + (float)k * ref.frac) / (float)VN, &sn, &cs);
                return mk(cs, sn);
            };
            cpx ph = phase_of(W * lt + s);
            const cpx step = phase_of(64 * W);
            float low = 0.0f;
#pragma unroll
            for (int i = 0; i < NLK; ++i) {
                const int k = W * (lt + 64 * i) + s;
                if (k < sq.nlow) {
                    const cpx x2 = L.xlow[lt + 64 * i];
                    const float pr = ph.x * lk_s[i].x - ph.y * lk_s[i].y;
                    const float pi = ph.x * lk_s[i].y + ph.y * lk_s[i].x;
                    const float rr = 0.5f * x2.x - ref.amp * pr;
                    const float ri = 0.5f * x2.y - ref.amp * pi;
                    low += ((k == 0) ? 1.0f : 2.0f) * lk_g[i] * (rr * rr + ri * ri);
                }
                ph = cmul(ph, step);
            }
            low = ofx_wave_sum(low);
            if (lane == 0) X.lowp[q][s] = low;
            __syncthreads();                            // the two partial sums; the scans of the dump are done
            if (s == 0 && lane == 0) {
                float lw = 0.0f;
#pragma unroll
                for (int t = 0; t < W; ++t) lw += X.lowp[q][t];
                ofx_write_search(row, sq, sd, pd.inv_fs, pre, chi0, best, lw, refine ? &ref : nullptr);
            }
        }
        vcur = vnext;
    }
}

}  // namespace

bool ofx_wave2_supported(int n_samples) { return n_samples == 8192 || n_samples == 16384; }

template <int W>
static int wave2_tables(ofx_plan* p) {
    if (p->d_tw1) return OFX_OK;
    constexpr int VN = 4096 * W, VM = 2048 * W;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<float2> t1(6 * 64), t2(128 + W * 128 + W * 64);
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
    for (int s = 0; s < W; ++s)
        for (int n = 0; n < 128; ++n) {     // ua[s][h][lane] = w_M'^{(lane + 64 h) s}
            const double a = -PI2 * (double)((n * s) % VM) / VM;
            t2[128 + s * 128 + n] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int s = 0; s < W; ++s)
        for (int v = 0; v < 64; ++v) {      // tbase[s][v] = i exp(-2 pi i (W v + s) / N)
            const double a = -PI2 * (double)(W * v + s) / VN;
            t2[128 + W * 128 + 64 * s + v] = make_float2((float)-std::sin(a), (float)std::cos(a));
        }
    return fused_upload_tables(p, t1, t2);
}

// Middle-step tables of one slot: per wave s the rows of ofx_wave.hip at the global bins k = W q + s,
// q = v + 128 j (wave 0, lane 0: 128 j for j < 8, 64 + 128 (j - 8) above), partner M' - k.
//   d_pq (float4 units): [s][16][64] midW, then [s][16][64] midG as float2, last entry the self-paired bin.
template <int W>
static int wave2_prepare_slot(ofx_plan* p, int slot, const double* wf) {
    int rc = wave2_tables<W>(p);
    if (rc) return rc;
    constexpr int VM = 2048 * W, NW = W * 16 * 64, NG = NW / 2;
    std::vector<float4> tab(NW + NG + 1, make_float4(0.f, 0.f, 0.f, 0.f));
    float2* tg = reinterpret_cast<float2*>(tab.data() + NW);
    for (int s = 0; s < W; ++s)
        fused_fill_slot_tables(tab.data() + s * 16 * 64, tg + s * 16 * 64, wf, p->slot[slot].g_host, VM, 64, 64, 16,
                               [s](int v, int j) {
                                   const int q = (s != 0 || v != 0) ? v + 128 * j
                                                                    : (j < 8 ? 128 * j : 64 + 128 * (j - 8));
                                   return W * q + s;
                               });
    return fused_finish_slot_tables(p, slot, wf, tab, VM);
}
int ofx_wave2_prepare_slot(ofx_plan* p, int slot, const double* wf) {
    return p->N == 8192 ? wave2_prepare_slot<2>(p, slot, wf) : wave2_prepare_slot<4>(p, slot, wf);
}

template <int W, int FEAT>
static int launch_wave2(ofx_plan* p, const OfxPlanDev& pd, const OfxSlotDev& sd, const Wave2Tabs& tabs,
                        const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                        hipStream_t st) {
    OFX_LDS_ATTR_ONCE((k_wave2<W, FEAT>), sizeof(Wave2Shared<W>));
    long long grid = (long long)p->cu_count * (8 / W);
    if (grid > n) grid = n;
    size_t tix = 0;
    int rc = ofx_time_begin(p, st, &tix);
    if (rc) return rc;
    hipLaunchKernelGGL((k_wave2<W, FEAT>), dim3((unsigned)grid), dim3(64 * W), sizeof(Wave2Shared<W>), st, pd, sd,
                       tabs, d_traces, d_valid, n, d_out);
    rc = ofx_time_end(p, st, tix);
    if (rc) return rc;
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

template <int W>
static int wave2_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                         hipStream_t st) {
    constexpr int VN = 4096 * W;
    int rc = wave2_tables<W>(p);
    if (rc) return rc;
    OfxPlanDev pd;
    ofx_fill_plan_dev(p, &pd);
    if constexpr (W == 2) {
        struct G { enum { N = 8192, ROWS = WROWS, NROWS = 32 }; };
        fused_classify_windows<G>(pd);
    }
    Wave2Tabs tabs;
    memset(&tabs, 0, sizeof(tabs));
    tabs.t1 = p->d_tw1;
    tabs.t2 = p->d_tw2;
    tabs.ua = p->d_tw2 + 128;
    tabs.tbase = p->d_tw2 + 128 + W * 128;
    {
        // wave 0, lane 0, slots j >= 8: bin W (64 + 128 (j - 8)) = 128 W j + W (64 - 1024):
        // i exp(-2 pi i W (64 - 1024) / N) = -exp(-2 pi i 64 / 4096)
        const double a = -6.283185307179586476925286766559 * 64.0 / 4096.0;
        tabs.tb0hi = make_float2((float)-std::cos(a), (float)-std::sin(a));
    }
    tabs.midW = reinterpret_cast<const float4*>(p->d_tw1);      // never read without searches
    tabs.midG = p->d_tw1;
    // Validate first, then one launch per filter slot with searches (as ofx_wave_process).
    int slots[OFX_MAX_SLOTS], nslots = 0;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s) {
        if (!p->slot[s].set || p->slot[s].searches.empty()) continue;
        OfxSlotDev sd;
        ofx_fill_slot_dev(p, s, &sd);
        for (int q = 0; q < sd.n_search; ++q)
            if (sd.search[q].nlow > W * WLOW) {
                ofx_set_error("FUSED engine (%d samples): lowchi2_fcutoff covers %d bins (> %d)", VN,
                              sd.search[q].nlow, W * WLOW);
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
            if (pd.band[i].k_hi > W * WLOW) {
                ofx_set_error("FUSED engine (%d samples): band [%d,%d) exceeds the %d stashed bins", VN,
                              pd.band[i].k_lo, pd.band[i].k_hi, W * WLOW);
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
            tabs.midG = reinterpret_cast<const float2*>(p->slot[s].d_pq + W * 16 * 64);
            tabs.wq = make_float2(p->slot[s].wq_x, p->slot[s].wq_y);
            tabs.gq = p->slot[s].gq;
        }
        if (li == 1) pd.n_tdwin = pd.n_bands = 0;
        int feat = 0;
        for (int q = 0; q < sd.n_search; ++q) {
            const OfxSearchDev& sq = sd.search[q];
            const bool full = sq.lo == 0 && sq.hi == VN && !sq.outside;
            if (sq.kind == OFX_SEARCH_DELAY && (sq.interp || !full)) feat |= 1;
        }
        if (pd.n_tdwin > 0) feat |= 2;
        if (p->n_channels > 1 || p->n_terms > 1 || p->weight[0] != 1.0) feat |= 4;
        switch (feat) {
#define OFX_CASE(F) case F: rc = launch_wave2<W, F>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st); break;
            OFX_CASE(0) OFX_CASE(1) OFX_CASE(2) OFX_CASE(3) OFX_CASE(4) OFX_CASE(5) OFX_CASE(6)
#undef OFX_CASE
            default: rc = launch_wave2<W, 7>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st);
        }
        if (rc) return rc;
    }
    return OFX_OK;
}
int ofx_wave2_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                      hipStream_t st) {
    return p->N == 8192 ? wave2_process<2>(p, d_traces, d_valid, n, d_out, st)
                        : wave2_process<4>(p, d_traces, d_valid, n, d_out, st);
}
