// ofx_lds.hip -- LDS engine: one workgroup carries a whole trace through
//   load -> forward FFT -> filter + chi2_0 -> inverse FFT -> searches -> one output row
// with the trace resident in LDS, for ANY even length N whose packed length M = N/2 has
// only the prime factors 2, 3 and 5 and fits in LDS (8 M bytes <= 136 KiB, N <= 34816):
// every sample is read from HBM once, like the FUSED engine, without being tied to
// N = 32768 (the reference's own example uses 25000-sample traces).
//
// The transform is an in-place mixed-radix (5, 4, 3, 2) decimation-in-frequency FFT of the
// packed points z[m] = x[2m] + i x[2m+1]; its output is digit-reversed, which the middle step
// (real-FFT unpack, filter, chi2_0, re-pack: the algebra of k_mid in ofx_rocfft.hip) absorbs
// by addressing bins through pos(k); the inverse is the exact mirror (conjugate twiddle, then
// inverse radix butterfly, stages in reverse order) and leaves A(2m) + i A(2m+1) in natural
// order, so the searches scan LDS.  Twiddles come from one table exp(-2 pi i j / N), j < N
// (L2-resident).  LDS-bound: 2 x (#stages) passes over the trace in LDS per slot.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ofx_common.h"
#include "ofx_device.h"
#include "ofx_fft_regs.h"

using ofxfft::cpx;
using ofxfft::mk;
using ofxfft::swp;
using ofxfft::pfma;
using ofxfft::cmul;
using ofxfft::cmulc;

namespace {

constexpr int LDS_MAX_FAC = 16;
constexpr int LDS_VLOW = 1024;           // low bins of V kept in LDS (lowchi2 / psd_amp)
constexpr size_t LDS_BUDGET = 150 * 1024;

struct LdsGeom {
    int M, N, nfac;
    int fac[LDS_MAX_FAC];
};

struct LdsSlot {
    OfxSlotDev sd;
};

// x * (-+ i): forward uses -i (DIR = -1), inverse +i
template <int DIR>
__device__ __forceinline__ cpx mul_i(cpx v) {
    return swp(v) * mk((float)-DIR, (float)DIR);     // DIR=+1: (-y, x) = i v ; DIR=-1: (y, -x) = -i v
}

// in-place radix-R DFT of x[0..R) with sign DIR (exp(DIR 2 pi i t q / R))
template <int R, int DIR>
__device__ __forceinline__ void dft_small(cpx (&x)[5]) {
    if constexpr (R == 2) {
        const cpx a = x[0] + x[1], b = x[0] - x[1];
        x[0] = a; x[1] = b;
    } else if constexpr (R == 4) {
        const cpx a = x[0] + x[2], b = x[0] - x[2], c = x[1] + x[3];
        const cpx d = mul_i<DIR>(x[1] - x[3]);
        x[0] = a + c; x[2] = a - c; x[1] = b + d; x[3] = b - d;
    } else if constexpr (R == 3) {
        constexpr float h = 0.86602540378443864676f;
        const cpx t1 = x[1] + x[2];
        const cpx m = pfma(t1, mk(-0.5f, -0.5f), x[0]);
        const cpx n = mul_i<DIR>(x[1] - x[2]) * mk(h, h);
        x[0] = x[0] + t1; x[1] = m + n; x[2] = m - n;
    } else {   // R == 5
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const cpx a1 = x[1] + x[4], a2 = x[2] + x[3], b1 = x[1] - x[4], b2 = x[2] - x[3];
        const cpx m1 = pfma(a2, mk(c2, c2), pfma(a1, mk(c1, c1), x[0]));
        const cpx m2 = pfma(a2, mk(c1, c1), pfma(a1, mk(c2, c2), x[0]));
        const cpx n1 = mul_i<DIR>(pfma(b2, mk(s2, s2), b1 * mk(s1, s1)));
        const cpx n2 = mul_i<DIR>(pfma(b2, mk(-s1, -s1), b1 * mk(s2, s2)));
        x[0] = x[0] + a1 + a2;
        x[1] = m1 + n1; x[4] = m1 - n1;
        x[2] = m2 + n2; x[3] = m2 - n2;
    }
}

// One stage over the whole array.  L: block length of the stage, r: radix, stride = L / r.
// Forward (DIF): butterfly, then output q times W_L^{j q}.  Inverse: input q times
// conj(W_L^{j q}), then inverse butterfly.  W_L^{e} = tw[e * (N / L)], e < L.
template <int R, bool FWD>
__device__ __forceinline__ void stage(cpx* z, const float2* __restrict__ tw, int M, int N, int L) {
    const int stride = L / R;
    const int tws = N / L;
    const int nbf = M / R;
    for (int b = threadIdx.x; b < nbf; b += blockDim.x) {
        const int blk = b / stride;
        const int j = b - blk * stride;
        cpx* base = z + blk * L + j;
        cpx x[5];
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = base[t * stride];
        if constexpr (!FWD) {
#pragma unroll
            for (int q = 1; q < R; ++q) {
                const float2 w = tw[(size_t)j * q * tws];
                x[q] = cmulc(x[q], mk(w.x, w.y));
            }
        }
        dft_small<R, FWD ? -1 : 1>(x);
        if constexpr (FWD) {
#pragma unroll
            for (int q = 1; q < R; ++q) {
                const float2 w = tw[(size_t)j * q * tws];
                x[q] = cmul(x[q], mk(w.x, w.y));
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) base[t * stride] = x[t];
    }
}

template <bool FWD>
__device__ __forceinline__ void stage_any(int r, cpx* z, const float2* tw, int M, int N, int L) {
    if (r == 5) stage<5, FWD>(z, tw, M, N, L);
    else if (r == 4) stage<4, FWD>(z, tw, M, N, L);
    else if (r == 3) stage<3, FWD>(z, tw, M, N, L);
    else stage<2, FWD>(z, tw, M, N, L);
}

// position of frequency bin k in the digit-reversed output of the forward transform
__device__ __forceinline__ int pos_of(int k, const LdsGeom& g) {
    int rem = k, p = 0, L = g.M;
    for (int i = 0; i < g.nfac; ++i) {
        const int r = g.fac[i];
        const int q = rem % r;
        rem /= r;
        L /= r;
        p += q * L;
    }
    return p;
}

template <int BT>
__global__ __launch_bounds__(BT) void k_lds(OfxPlanDev pd, LdsGeom g, const LdsSlot* __restrict__ slots,
                                            int nslots, const float2* __restrict__ tw,
                                            const float* __restrict__ traces,
                                            const uint8_t* __restrict__ valid, long long n_traces,
                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* z = reinterpret_cast<cpx*>(smem);                         // [M]
    cpx* vlow = z + g.M;                                           // [LDS_VLOW]
    float* scratch = reinterpret_cast<float*>(vlow + LDS_VLOW);    // [BT / 64]
    OfxCand* cscratch = reinterpret_cast<OfxCand*>(scratch + 32);  // [BT / 64]
    const int tid = threadIdx.x;
    const int M = g.M, N = g.N, pre = pd.pre;
    const float* a = reinterpret_cast<const float*>(z);            // lags after the inverse

    for (long long b = blockIdx.x; b < n_traces; b += gridDim.x) {
        float* row = out + (size_t)b * pd.row;
        if (valid && !valid[b]) {
            for (int j = tid; j < pd.row; j += BT) row[j] = OFX_SENTINEL;
            continue;
        }
        const float* e = traces + (size_t)b * pd.n_channels * N;
        const bool plain = (pd.n_terms == 1 && pd.weight[0] == 1.0f);
        auto sample2 = [&](int m) -> cpx {
            if (plain) {
                const float2 v = reinterpret_cast<const float2*>(e + (size_t)pd.chan[0] * N)[m];
                return mk(v.x, v.y);
            }
            cpx acc = mk(0.f, 0.f);
            for (int c = 0; c < pd.n_terms; ++c) {
                const float2 v = reinterpret_cast<const float2*>(e + (size_t)pd.chan[c] * N)[m];
                acc = pfma(mk(pd.weight[c], pd.weight[c]), mk(v.x, v.y), acc);
            }
            return acc;
        };
        const int pass_n = (nslots > 0) ? nslots : 1;
        for (int si = 0; si < pass_n; ++si) {
            __syncthreads();                       // previous readers of z are done
            for (int m = tid; m < M; m += BT) z[m] = sample2(m);
            __syncthreads();
            // ---------------------------------------------- time-domain windows (once)
            if (si == 0) {
                for (int w = 0; w < pd.n_tdwin; ++w) {
                    const int lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
                    float s = 0.0f, sq = 0.0f, mx = -INFINITY, mn = INFINITY;
                    for (int n = lo + tid; n < hi; n += BT) {
                        const float v = a[n];
                        s += v;
                        sq = fmaf(v, v, sq);
                        mx = fmaxf(mx, v);
                        mn = fminf(mn, v);
                    }
                    s = ofx_block_sum(s, scratch);
                    sq = ofx_block_sum(sq, scratch);
                    mx = ofx_block_max(mx, scratch);
                    mn = ofx_block_min(mn, scratch);
                    if (tid == 0) {
                        const float first = a[lo], last = a[hi - 1];
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
                if (nslots == 0 && pd.n_bands == 0) break;
                __syncthreads();
            }
            // -------------------------------------------------------------- forward
            {
                int L = M;
                for (int i = 0; i < g.nfac; ++i) {
                    stage_any<true>(g.fac[i], z, tw, M, N, L);
                    L /= g.fac[i];
                    __syncthreads();
                }
            }
            // ------------------------------------------------ middle (k_mid algebra)
            const OfxSlotDev* sdp = (nslots > 0) ? &slots[si].sd : nullptr;
            const float2* wf = sdp ? sdp->wf : nullptr;
            const float* gg = sdp ? sdp->g : nullptr;
            float acc = 0.0f;
            for (int k = tid; k <= M / 2; k += BT) {
                const int p = (k == 0) ? 0 : M - k;
                const int pk = pos_of(k, g), pp = (p == k) ? pk : pos_of(p, g);
                const cpx zk = z[pk], zp = z[pp];
                const float2 t2 = tw[k];                                  // t_k = exp(-2 pi i k / N)
                const float cs = t2.x, sn = t2.y;
                cpx vk, vpc;
                if (k == 0) {
                    vk = mk(zk.x + zk.y, 0.0f);
                    vpc = mk(zk.x - zk.y, 0.0f);
                } else {
                    const cpx u = mk(zk.x + zp.x, zk.y - zp.y);
                    const cpx w = mk(zk.x - zp.x, zk.y + zp.y);
                    const cpx sv = mk(-(cs * w.y + sn * w.x), cs * w.x - sn * w.y);
                    vk = (u - sv) * mk(0.5f, 0.5f);
                    vpc = (u + sv) * mk(0.5f, 0.5f);
                }
                const int kp = (k == 0) ? M : p;
                if (k < LDS_VLOW) vlow[k] = vk;
                if (kp < LDS_VLOW && kp != k) vlow[kp] = mk(vpc.x, -vpc.y);
                if (!wf) continue;
                const float wk = (k == 0) ? 1.0f : 2.0f;
                const float wp = (kp == M) ? 1.0f : 2.0f;
                acc = fmaf(wk * gg[k], vk.x * vk.x + vk.y * vk.y, acc);
                if (kp != k) acc = fmaf(wp * gg[kp], vpc.x * vpc.x + vpc.y * vpc.y, acc);
                const float2 fa = wf[k], fc = wf[kp];
                const cpx yk = mk(fa.x * vk.x - fa.y * vk.y, fa.x * vk.y + fa.y * vk.x);
                const cpx ypc = mk(fc.x * vpc.x + fc.y * vpc.y, fc.x * vpc.y - fc.y * vpc.x);
                const cpx ye = yk + ypc, d = yk - ypc;
                const cpx yo = mk(d.x * cs + d.y * sn, d.y * cs - d.x * sn);
                z[pk] = mk(ye.x - yo.y, ye.y + yo.x);
                if (p != k) z[pp] = mk(ye.x + yo.y, -(ye.y - yo.x));
            }
            __syncthreads();
            // psd_amp bands (first pass only)
            if (si == 0 && pd.n_bands > 0) {
                const float c = 1.0f / ((float)N * pd.fs);
                for (int i = 0; i < pd.n_bands; ++i) {
                    const int lo = pd.band[i].k_lo, hi = pd.band[i].k_hi;
                    float accb = 0.0f;
                    for (int k = lo + tid; k < hi; k += BT) {
                        const cpx v = vlow[k];
                        const float w = (2 * k == N) ? 1.0f : 2.0f;
                        accb += sqrtf(w * c * (v.x * v.x + v.y * v.y));
                    }
                    accb = ofx_block_sum(accb, scratch);
                    if (tid == 0) row[pd.band[i].out_off] = accb / (float)(hi - lo);
                }
            }
            if (!sdp) break;
            const float chi0 = ofx_block_sum(acc, scratch);
            // -------------------------------------------------------------- inverse
            {
                int Ls[LDS_MAX_FAC];
                int L = M;
                for (int i = 0; i < g.nfac; ++i) {
                    Ls[i] = L;
                    L /= g.fac[i];
                }
                for (int i = g.nfac - 1; i >= 0; --i) {
                    stage_any<false>(g.fac[i], z, tw, M, N, Ls[i]);
                    __syncthreads();
                }
            }
            // ------------------------------------------------------------- searches
            const OfxSlotDev& sd = *sdp;
            for (int q = 0; q < sd.n_search; ++q) {
                const OfxSearchDev sq = sd.search[q];
                OfxCand best = ofx_cand_none();
                auto scan = [&](int i0, int i1) {
                    for (int i = i0 + tid; i < i1; i += BT) {
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
                if (sq.kind == OFX_SEARCH_NODELAY) {
                    scan(pre, pre + 1);
                } else if (sq.outside) {
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
                ref.chi2 = 0.0f;
                if (sq.interp) {
                    float am = 0.f, ap = 0.f;
                    if (best.idx > 0 && best.idx < N - 1) {
                        int nm = best.idx - 1 - pre, np = best.idx + 1 - pre;
                        if (nm < 0) nm += N;
                        if (np < 0) np += N;
                        am = a[nm];
                        ap = a[np];
                    }
                    ref = ofx_interpolate(am, best.amp, ap, best.idx, N, sd.norm, chi0);
                }
                float low = 0.0f;
                for (int k = tid; k < sq.nlow; k += BT) {
                    const cpx v = vlow[k];
                    low += ofx_lowchi2_term(k, N, d, ref.amp, make_float2(v.x, v.y), sd.s[k],
                                            sd.g[k], ref.frac);
                }
                low = ofx_block_sum(low, scratch);
                if (tid == 0)
                    ofx_write_search(row, sq, sd, pd.inv_fs, pre, chi0, best, low,
                                     sq.interp ? &ref : nullptr);
            }
        }
    }
}

bool factorize(int M, std::vector<int>* fac) {
    fac->clear();
    int rem = M;
    while (rem % 5 == 0) { fac->push_back(5); rem /= 5; }
    while (rem % 4 == 0) { fac->push_back(4); rem /= 4; }
    while (rem % 3 == 0) { fac->push_back(3); rem /= 3; }
    while (rem % 2 == 0) { fac->push_back(2); rem /= 2; }
    return rem == 1 && (int)fac->size() <= LDS_MAX_FAC;
}

size_t lds_bytes_for(int M) {
    return (size_t)M * 8 + (size_t)LDS_VLOW * 8 + 32 * 4 + 32 * sizeof(OfxCand);
}

}  // namespace

bool ofx_lds_supported(int n_samples) {
    if (n_samples < 16 || (n_samples % 2)) return false;
    std::vector<int> fac;
    if (!factorize(n_samples / 2, &fac)) return false;
    return lds_bytes_for(n_samples / 2) <= LDS_BUDGET;
}

int ofx_lds_release(ofx_plan* p) {
    if (p->d_lds_tw) (void)hipFree(p->d_lds_tw);
    if (p->d_lds_slots) (void)hipFree(p->d_lds_slots);
    p->d_lds_tw = nullptr;
    p->d_lds_slots = nullptr;
    return OFX_OK;
}

template <int BT>
static int launch_lds(ofx_plan* p, const OfxPlanDev& pd, const LdsGeom& g, int nslots,
                      const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                      hipStream_t st) {
    const size_t lds = lds_bytes_for(g.M);
    static size_t attr_done = 0;
    if (attr_done < lds) {
        OFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<BT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
        attr_done = LDS_BUDGET;
    }
    int per_cu = (int)((160 * 1024) / lds);
    const int by_threads = 2048 / BT;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)p->cu_count * per_cu;
    if (grid > n) grid = n;
    size_t tix = 0;
    int rc = ofx_time_begin(p, st, &tix);
    if (rc) return rc;
    hipLaunchKernelGGL(k_lds<BT>, dim3((unsigned)grid), dim3(BT), lds, st, pd, g,
                       reinterpret_cast<const LdsSlot*>(p->d_lds_slots), nslots,
                       reinterpret_cast<const float2*>(p->d_lds_tw), d_traces, d_valid, n, d_out);
    rc = ofx_time_end(p, st, tix);
    if (rc) return rc;
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

int ofx_lds_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n,
                    float* d_out, hipStream_t st) {
    const int N = p->N, M = N / 2;
    std::vector<int> fac;
    if (!factorize(M, &fac) || lds_bytes_for(M) > LDS_BUDGET) {
        ofx_set_error("LDS engine: n_samples=%d is not supported", N);
        return OFX_ERR_UNSUPPORTED;
    }
    OfxPlanDev pd;
    ofx_fill_plan_dev(p, &pd);
    LdsGeom g;
    memset(&g, 0, sizeof(g));
    g.M = M;
    g.N = N;
    g.nfac = (int)fac.size();
    for (int i = 0; i < g.nfac; ++i) g.fac[i] = fac[i];
    if (!p->d_lds_tw) {
        std::vector<float2> tw(N);
        const double c = -6.283185307179586476925286766559 / (double)N;
        for (int j = 0; j < N; ++j) tw[j] = make_float2((float)std::cos(c * j), (float)std::sin(c * j));
        OFX_HIP(hipMalloc(&p->d_lds_tw, sizeof(float2) * (size_t)N));
        OFX_HIP(hipMemcpy(p->d_lds_tw, tw.data(), sizeof(float2) * (size_t)N, hipMemcpyHostToDevice));
    }
    std::vector<LdsSlot> args;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s) {
        if (!p->slot[s].set || p->slot[s].searches.empty()) continue;
        LdsSlot a;
        memset(&a, 0, sizeof(a));
        ofx_fill_slot_dev(p, s, &a.sd);
        for (int q = 0; q < a.sd.n_search; ++q)
            if (a.sd.search[q].nlow > LDS_VLOW) {
                ofx_set_error("LDS engine: lowchi2_fcutoff covers %d bins (> %d)",
                              a.sd.search[q].nlow, LDS_VLOW);
                return OFX_ERR_UNSUPPORTED;
            }
        args.push_back(a);
    }
    for (const auto& bd : p->bands)
        if (bd.k_hi > LDS_VLOW) {
            ofx_set_error("LDS engine: band [%d,%d) exceeds the %d stashed bins", bd.k_lo, bd.k_hi,
                          LDS_VLOW);
            return OFX_ERR_UNSUPPORTED;
        }
    const int nslots = (int)args.size();
    if (nslots > 0) {
        if (!p->d_lds_slots) OFX_HIP(hipMalloc(&p->d_lds_slots, sizeof(LdsSlot) * OFX_MAX_SLOTS));
        OFX_HIP(hipMemcpyAsync(p->d_lds_slots, args.data(), sizeof(LdsSlot) * (size_t)nslots,
                               hipMemcpyHostToDevice, st));
    }
    if (M >= 8192) return launch_lds<512>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st);
    return launch_lds<256>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st);
}
