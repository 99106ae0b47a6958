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
// (L2-resident).  LDS-bound: 2 x (#stages) passes over the trace in LDS per slot (the
// forward half only once when a plan has several slots).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ofx_common.h"
#include <cstdlib>

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
constexpr size_t LDS_BUDGET = 160 * 1024;
constexpr size_t LDS_FIXED_BYTES = LDS_VLOW * 8 + 32 * 4 + 32 * 16;   // besides data and twiddles

struct LdsGeom {
    int M, N, nfac;
    int fac[LDS_MAX_FAC];
    unsigned magic[LDS_MAX_FAC];   // floor(2^32 / stride) + 1 per stage: b / stride = umulhi(b, magic)
    int toff[LDS_MAX_FAC];         // offset of the stage's twiddles W_L^j, j < stride, in the LDS table
    int vlow;                      // low bins of V kept in LDS (a multiple of 64, <= LDS_VLOW)
    int anch[LDS_MAX_FAC];         // twiddle tables of the stage (radix 8 / 16): 1 = W^j only,
                                   // 2 = W^j, W^{4j}, 3 = W^{qj} for q = 1..4 (and 8, 12)
    int ntw;                       // entries of that table (tables x strides > 1)
};

// middle-step constants of one bin pair (k, M - k), 48 bytes, built per slot on the host
struct __attribute__((aligned(16))) LdsPair {
    int pk, pp;            // positions of the two bins in the digit-reversed spectrum
    float tx, ty;          // t_k = exp(-2 pi i k / N)
    float wkx, wky;        // wf_k
    float wpx, wpy;        // wf_p   (p = M - k; k = 0: the Nyquist bin)
    float gk, gp;          // chi2 weights w g of the two bins (0 if the pair is one bin)
    int k;                 // the bin (pairs are stored in order of pk, not of k: consecutive
    float pad1;            // lanes then touch consecutive-ish LDS positions)
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
__device__ __forceinline__ void dft_small(cpx (&x)[16]) {
    if constexpr (R == 16 || R == 8) {
        ofxfft::dft<R, DIR, 16, 0>(x);
    } else if constexpr (R == 2) {
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
// conj(W_L^{j q}), then inverse butterfly.  The powers come from per-stage tables copied into
// LDS once per workgroup.  A chain of products from W_L^j alone multiplies that entry's fp32
// rounding error by q (up to 15 in a radix-16 stage: chi2 of 4096-sample traces was off by
// up to 2.8e-5 relative), so radix-8 / 16 stages carry more tables where LDS permits
// (lds_layout): T = 2 adds W^{4j} (error <= 3 + 3 entry roundings), T = 3 tabulates
// q = 1..4 (and 8, 12): every power is one table entry or one product of two.
template <int R>
__device__ __forceinline__ void twiddle_powers(cpx (&w)[16], const cpx* tw1, int j, int stride,
                                               int T) {
    w[1] = tw1[j];
    if constexpr (R >= 8) {
        if (T == 4) {                          // every power from its own table
#pragma unroll
            for (int q = 2; q < R; ++q) w[q] = tw1[(q - 1) * stride + j];
            return;
        }
        if (T >= 2) {
            if (T == 3) {
                w[2] = tw1[stride + j];
                w[3] = tw1[2 * stride + j];
                w[4] = tw1[3 * stride + j];
            } else {
                w[4] = tw1[stride + j];
                w[2] = cmul(w[1], w[1]);
                w[3] = cmul(w[2], w[1]);
            }
            w[5] = cmul(w[4], w[1]);
            w[6] = cmul(w[4], w[2]);
            w[7] = cmul(w[4], w[3]);
            if constexpr (R > 8) {
                if (T == 3) {
                    w[8] = tw1[4 * stride + j];
                    w[12] = tw1[5 * stride + j];
                } else {
                    w[8] = cmul(w[4], w[4]);
                    w[12] = cmul(w[8], w[4]);
                }
                w[9] = cmul(w[8], w[1]);
                w[10] = cmul(w[8], w[2]);
                w[11] = cmul(w[8], w[3]);
                w[13] = cmul(w[12], w[1]);
                w[14] = cmul(w[12], w[2]);
                w[15] = cmul(w[12], w[3]);
            }
            return;
        }
    }
    if constexpr (R > 2) w[2] = cmul(w[1], w[1]);
    if constexpr (R > 3) w[3] = cmul(w[2], w[1]);
    if constexpr (R > 4) {
        w[4] = cmul(w[2], w[2]);
    }
    if constexpr (R > 5) {
        w[5] = cmul(w[4], w[1]);
        w[6] = cmul(w[4], w[2]);
        w[7] = cmul(w[4], w[3]);
    }
    if constexpr (R > 8) {
        w[8] = cmul(w[4], w[4]);
        w[9] = cmul(w[8], w[1]);
        w[10] = cmul(w[8], w[2]);
        w[11] = cmul(w[8], w[3]);
        w[12] = cmul(w[8], w[4]);
        w[13] = cmul(w[12], w[1]);
        w[14] = cmul(w[12], w[2]);
        w[15] = cmul(w[12], w[3]);
    }
}

template <int R, bool FWD>
__device__ __forceinline__ void stage(cpx* z, const cpx* tw1, int M, int L,
                                      unsigned magic, int T) {
    const int stride = L / R;
    const int nbf = M / R;
    for (int b = threadIdx.x; b < nbf; b += blockDim.x) {
        // exact for b, stride < 2^16 (M <= 17408)
        const int blk = (stride == 1) ? b : (int)__umulhi((unsigned)b, magic);
        const int j = b - blk * stride;
        cpx* base = z + blk * L + j;
        cpx x[16], w[16];
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = base[t * stride];
        const bool tw_needed = (stride > 1);          // last stage: every twiddle is 1
        if (tw_needed) twiddle_powers<R>(w, tw1, j, stride, T);
        if constexpr (!FWD) {
            if (tw_needed) {
#pragma unroll
                for (int q = 1; q < R; ++q) x[q] = cmulc(x[q], w[q]);
            }
        }
        dft_small<R, FWD ? -1 : 1>(x);
        if constexpr (FWD) {
            if (tw_needed) {
#pragma unroll
                for (int q = 1; q < R; ++q) x[q] = cmul(x[q], w[q]);
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) base[t * stride] = x[t];
    }
}

// MAXR: only radices <= MAXR are compiled in (a kernel for lengths without a radix-16 stage,
// or without radix 8 and 16, needs far fewer registers and can run with 1024 threads)
template <bool FWD, int MAXR = 16>
__device__ __forceinline__ void stage_any(int r, cpx* z, const cpx* tw1, int M, int L,
                                          unsigned magic, int T) {
    if constexpr (MAXR >= 16) {
        if (r == 16) { stage<16, FWD>(z, tw1, M, L, magic, T); return; }
    }
    if constexpr (MAXR >= 8) {
        if (r == 8) { stage<8, FWD>(z, tw1, M, L, magic, T); return; }
    }
    if (r == 5) stage<5, FWD>(z, tw1, M, L, magic, 1);
    else if (r == 4) stage<4, FWD>(z, tw1, M, L, magic, 1);
    else if (r == 3) stage<3, FWD>(z, tw1, M, L, magic, 1);
    else stage<2, FWD>(z, tw1, M, L, magic, 1);
}

// position of frequency bin k in the digit-reversed output of the forward transform
// (tabulated once per plan on the host)
static int pos_of(int k, const LdsGeom& g) {
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

// Batched complex transform of length M with natural-order rows in global memory (the
// transform of k_lds on its own): forward reads a row into LDS, runs the DIF stages and gathers
// the bins from their digit-reversed positions; inverse scatters the bins to those positions,
// runs the DIT stages and writes the row.  Unnormalised, as rocFFT.  Used by the N x M engine
// for trace lengths where rocFFT falls back to its multi-kernel path.
template <int BT, bool FWD, int MAXR = 16>
__global__ __launch_bounds__(BT) void k_lds_fft(LdsGeom g, const float2* __restrict__ stw,
                                                const int* __restrict__ pos,
                                                const float2* __restrict__ in,
                                                float2* __restrict__ out, long long rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* z = reinterpret_cast<cpx*>(smem);                         // [M]
    cpx* tw1 = z + g.M;                                            // [ntw]
    const int tid = threadIdx.x, M = g.M;
    for (int i = tid; i < g.ntw; i += BT) tw1[i] = mk(stw[i].x, stw[i].y);
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float2* src = in + (size_t)r * M;
        float2* dst = out + (size_t)r * M;
        __syncthreads();                                           // readers of the last row
        for (int m0 = tid; m0 < M; m0 += 8 * BT) {                 // batches of independent loads
            float2 tmp[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (m0 + i * BT < M) tmp[i] = src[m0 + i * BT];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (m0 + i * BT < M) {
                    const int m = m0 + i * BT;
                    z[FWD ? m : pos[m]] = mk(tmp[i].x, tmp[i].y);
                }
        }
        __syncthreads();
        if constexpr (FWD) {
            int L = M;
            for (int i = 0; i < g.nfac; ++i) {
                stage_any<true, MAXR>(g.fac[i], z, tw1 + g.toff[i], M, L, g.magic[i], g.anch[i]);
                L /= g.fac[i];
                __syncthreads();
            }
        } else {
            int Ls[LDS_MAX_FAC];
            int L = M;
            for (int i = 0; i < g.nfac; ++i) {
                Ls[i] = L;
                L /= g.fac[i];
            }
            for (int i = g.nfac - 1; i >= 0; --i) {
                stage_any<false, MAXR>(g.fac[i], z, tw1 + g.toff[i], M, Ls[i], g.magic[i], g.anch[i]);
                __syncthreads();
            }
        }
        for (int m = tid; m < M; m += BT) {
            const cpx v = z[FWD ? pos[m] : m];
            dst[m] = make_float2(v.x, v.y);
        }
    }
}

template <int BT, bool PF, int MAXR = 16>
__global__ __launch_bounds__(BT) void k_lds(OfxPlanDev pd, LdsGeom g, const LdsSlot* __restrict__ slots,
                                            int nslots, const float2* __restrict__ stw,
                                            const LdsPair* __restrict__ pairs,
                                            const float* __restrict__ traces,
                                            const uint8_t* __restrict__ valid, long long n_traces,
                                            float* __restrict__ out, float2* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* z = reinterpret_cast<cpx*>(smem);                         // [M]
    cpx* vlow = z + g.M;                                           // [g.vlow]
    cpx* tw1 = vlow + g.vlow;                                      // [ntw] W_L^j per stage
    float* scratch = reinterpret_cast<float*>(tw1 + g.ntw);        // [BT / 64]
    OfxCand* cscratch = reinterpret_cast<OfxCand*>(scratch + 32);  // [BT / 64]
    const int tid = threadIdx.x;
    const int M = g.M, N = g.N, pre = pd.pre;
    const float* a = reinterpret_cast<const float*>(z);            // lags after the inverse
    for (int i = tid; i < g.ntw; i += BT) tw1[i] = mk(stw[i].x, stw[i].y);

    // Software-pipelined trace load (when a trace fits NPF values per thread): the next
    // trace is requested into registers before the searches of the current one, so its HBM
    // latency hides under them -- with one workgroup per CU nothing else would.
    constexpr int NPF = PF ? (BT >= 1024 ? 17 : 34) : 1;     // NPF BT >= 17408 values
    cpx pf[NPF];
    bool have_pf = false;
    const bool can_pf = PF && (M <= NPF * BT);
    for (long long b = blockIdx.x; b < n_traces; b += gridDim.x) {
        float* row = out + (size_t)b * pd.row;
        if (valid && !valid[b]) {
            for (int j = tid; j < pd.row; j += BT) row[j] = OFX_SENTINEL;
            have_pf = false;
            continue;
        }
        const bool plain = (pd.n_terms == 1 && pd.weight[0] == 1.0f);
        auto sample2 = [&](long long bb, int m) -> cpx {
            const float* e = traces + (size_t)bb * pd.n_channels * N;
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
        // several slots: the (digit-reversed) spectrum is parked in this workgroup's scratch
        // area after the first forward transform and read back for the other slots, so the
        // load and the forward transform are paid once per event
        float2* myspec = spec ? spec + (size_t)blockIdx.x * M : nullptr;
        for (int si = 0; si < pass_n; ++si) {
            __syncthreads();                       // previous readers of z are done
            if (si > 0) {
                for (int m = tid; m < M; m += BT) {
                    const float2 v = myspec[m];
                    z[m] = mk(v.x, v.y);
                }
            } else if (have_pf) {
                // the trace was requested during the previous event's searches
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    const int m = tid + i * BT;
                    if (m < M) z[m] = pf[i];
                }
            } else if constexpr (PF) {
                // cold start / extra slots: batches of independent loads, then the LDS writes
                for (int m0 = tid; m0 < M; m0 += 8 * BT) {
                    cpx tmp[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (m0 + i * BT < M) tmp[i] = sample2(b, m0 + i * BT);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (m0 + i * BT < M) z[m0 + i * BT] = tmp[i];
                }
            } else {
                for (int m = tid; m < M; m += BT) z[m] = sample2(b, m);
            }
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
                if (nslots == 0 && pd.n_bands == 0) {
                    have_pf = false;
                    break;
                }
                __syncthreads();
            }
            // -------------------------------------------------------------- forward
            if (si == 0) {
                int L = M;
                for (int i = 0; i < g.nfac; ++i) {
                    stage_any<true, MAXR>(g.fac[i], z, tw1 + g.toff[i], M, L, g.magic[i], g.anch[i]);
                    L /= g.fac[i];
                    __syncthreads();
                }
                if (pass_n > 1) {
                    for (int m = tid; m < M; m += BT) myspec[m] = make_float2(z[m].x, z[m].y);
                    __syncthreads();       // the middle step rewrites z in place, at other threads' m
                }
            } else {
                __syncthreads();
            }
            // ------------------------------------------------ middle (k_mid algebra)
            const OfxSlotDev* sdp = (nslots > 0) ? &slots[si].sd : nullptr;
            // pair table of this slot (slot 0's geometry part serves a plan without searches);
            // the next pair's 48 bytes are requested before the current pair is worked on
            const LdsPair* pt = pairs + (size_t)((nslots > 0) ? si : 0) * (M / 2 + 1);
            float acc = 0.0f;
            LdsPair cur;
            if (tid <= M / 2) cur = pt[tid];
#pragma unroll 2
            for (int it = tid; it <= M / 2; it += BT) {
                LdsPair nxt;
                if (it + BT <= M / 2) nxt = pt[it + BT];
                const int k = cur.k;
                const int p = (k == 0) ? 0 : M - k;
                const cpx zk = z[cur.pk], zp = z[cur.pp];
                const float cs = cur.tx, sn = cur.ty;
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
                if (k < g.vlow) vlow[k] = vk;
                if (kp < g.vlow && kp != k) vlow[kp] = mk(vpc.x, -vpc.y);
                if (sdp) {
                    acc = fmaf(cur.gk, vk.x * vk.x + vk.y * vk.y, acc);
                    acc = fmaf(cur.gp, vpc.x * vpc.x + vpc.y * vpc.y, acc);
                    const cpx yk = mk(cur.wkx * vk.x - cur.wky * vk.y, cur.wkx * vk.y + cur.wky * vk.x);
                    const cpx ypc = mk(cur.wpx * vpc.x + cur.wpy * vpc.y,
                                       cur.wpx * vpc.y - cur.wpy * vpc.x);
                    const cpx ye = yk + ypc, d = yk - ypc;
                    const cpx yo = mk(d.x * cs + d.y * sn, d.y * cs - d.x * sn);
                    z[cur.pk] = mk(ye.x - yo.y, ye.y + yo.x);
                    if (p != k) z[cur.pp] = mk(ye.x + yo.y, -(ye.y - yo.x));
                }
                cur = nxt;
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
            if (!sdp) {
                have_pf = false;
                break;
            }
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
                    stage_any<false, MAXR>(g.fac[i], z, tw1 + g.toff[i], M, Ls[i], g.magic[i], g.anch[i]);
                    __syncthreads();
                }
            }
            if (si == pass_n - 1) {
                const long long bn = b + gridDim.x;
                have_pf = can_pf && (bn < n_traces) && !(valid && !valid[bn]);
                if (have_pf) {
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int m = tid + i * BT;
                        if (m < M) pf[i] = sample2(bn, m);
                    }
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
    // even radices first, odd ones last: the late stages (small strides, lanes spread over
    // many blocks) then step through LDS with odd strides and stay free of bank conflicts
    int rem = M;
    while (rem % 16 == 0) { fac->push_back(16); rem /= 16; }
    while (rem % 8 == 0) { fac->push_back(8); rem /= 8; }
    while (rem % 4 == 0) { fac->push_back(4); rem /= 4; }
    while (rem % 2 == 0) { fac->push_back(2); rem /= 2; }
    while (rem % 5 == 0) { fac->push_back(5); rem /= 5; }
    while (rem % 3 == 0) { fac->push_back(3); rem /= 3; }
    return rem == 1 && (int)fac->size() <= LDS_MAX_FAC;
}

// tables of one stage: exponents q of the tabulated W_L^{q j}
int anchor_list(int r, int T, int (&e)[16]) {
    e[0] = 1;
    if (r < 8 || T <= 1) return 1;
    if (T == 4) {
        for (int q = 1; q < r; ++q) e[q - 1] = q;
        return r - 1;
    }
    if (T == 2) { e[1] = 4; return 2; }
    e[1] = 2; e[2] = 3; e[3] = 4;
    if (r == 8) return 4;
    e[4] = 8; e[5] = 12;
    return 6;
}

int stage_twiddle_count(int M, const std::vector<int>& fac, int T = 1) {
    int L = M, n = 0;
    for (int r : fac) {
        int e[16];
        if (L / r > 1) n += anchor_list(r, T, e) * (L / r);
        L /= r;
    }
    return n;
}

// How many twiddle tables the radix-8 / 16 stages of a length-M transform get: the highest
// level (3, then 2) that fits the LDS budget and still leaves three workgroups per CU (or as
// many as the single-table layout had, if fewer).  Level 3 is also the fastest where it fits
// (9 instead of 14 products per radix-16 butterfly: 4096 samples 38.5 -> 49.8 M traces/s against
// level 2) and brings the error at 4096 samples to that of the rocFFT path (chi2 max-rel
// 7.7e-6, no time-bin flip in 8192 events; level 1: 2.8e-5, level 2: 1.4e-5).
// other_bytes = everything else the kernel keeps in LDS besides data and twiddles.
int choose_anchor_level(int M, const std::vector<int>& fac, size_t other_bytes) {
    const size_t base = other_bytes + (size_t)M * 8;
    const size_t b1 = base + (size_t)stage_twiddle_count(M, fac, 1) * 8;
    if (const char* dg = getenv("OFX_DIAG_LDS_T")) {      // diagnostic: force the table level
        const int T = atoi(dg);
        if (T >= 1 && T <= 4 && base + (size_t)stage_twiddle_count(M, fac, T) * 8 <= LDS_BUDGET)
            return T;
    }
    const size_t want = std::min<size_t>(LDS_BUDGET / b1, 3);
    for (int T = 3; T >= 2; --T) {
        const size_t bt = base + (size_t)stage_twiddle_count(M, fac, T) * 8;
        if (bt <= LDS_BUDGET && LDS_BUDGET / bt >= want) return T;
    }
    return 1;
}

// stage geometry and twiddle tables (host copy) of a factorisation
void lds_layout(int M, const std::vector<int>& fac, int T, LdsGeom* g, std::vector<float2>* tab) {
    memset(g, 0, sizeof(*g));
    g->M = M;
    g->N = 2 * M;
    g->vlow = LDS_VLOW;
    g->nfac = (int)fac.size();
    int L = M;
    for (int i = 0; i < g->nfac; ++i) {
        g->fac[i] = fac[i];
        const unsigned stride = (unsigned)(L / fac[i]);
        g->magic[i] = (unsigned)((1ull << 32) / stride) + 1u;     // unused when stride == 1
        g->toff[i] = g->ntw;
        int e[16];
        const int nt = anchor_list(fac[i], T, e);
        g->anch[i] = (fac[i] >= 8) ? T : 1;
        if (stride > 1) g->ntw += nt * (int)stride;
        L /= fac[i];
    }
    if (!tab) return;
    tab->assign((size_t)std::max(1, g->ntw), make_float2(0.f, 0.f));
    L = M;
    for (int i = 0; i < g->nfac; ++i) {
        const int stride = L / g->fac[i];
        int e[16];
        const int nt = anchor_list(g->fac[i], T, e);
        if (stride > 1)
            for (int a = 0; a < nt; ++a)
                for (int j = 0; j < stride; ++j) {
                    const long long m = ((long long)j * e[a]) % L;                   // W_L^{e j}
                    const double ang = -6.283185307179586476925286766559 * (double)m / (double)L;
                    (*tab)[g->toff[i] + a * stride + j] =
                        make_float2((float)std::cos(ang), (float)std::sin(ang));
                }
        L /= g->fac[i];
    }
}

constexpr size_t LDS_OTHER_BYTES = (size_t)LDS_VLOW * 8 + 32 * 4 + 32 * sizeof(OfxCand);
size_t lds_bytes_for(int M, const std::vector<int>& fac) {
    const int T = choose_anchor_level(M, fac, LDS_OTHER_BYTES);
    return (size_t)M * 8 + LDS_OTHER_BYTES + (size_t)stage_twiddle_count(M, fac, T) * 8;
}

// no radix-16 stage: at most one more stage, but butterflies small enough for 1024 threads
bool factorize_small(int M, std::vector<int>* fac) {
    fac->clear();
    int rem = M;
    while (rem % 8 == 0) { fac->push_back(8); rem /= 8; }
    while (rem % 4 == 0) { fac->push_back(4); rem /= 4; }
    while (rem % 2 == 0) { fac->push_back(2); rem /= 2; }
    while (rem % 5 == 0) { fac->push_back(5); rem /= 5; }
    while (rem % 3 == 0) { fac->push_back(3); rem /= 3; }
    return rem == 1 && (int)fac->size() <= LDS_MAX_FAC;
}

// Factors of a length-M transform in LDS.  When the data alone takes more than half of the LDS
// (one workgroup per CU), the factorisation without a radix-16 stage is taken if its twiddle
// tables still fit: the kernels then run with 1024
// threads, whose sixteen waves hide the LDS latency of the stages.
bool choose_factors(int M, size_t fixed_bytes, std::vector<int>* fac, bool* small) {
    *small = false;
    const char* dg = getenv("OFX_DIAG_LDS_SMALL");
    // measured on MI355X: also pays for powers of two in that regime (32768 samples: 4.6 -> 5.3 M
    // traces/s) and at 8192 samples (8^4 points: 19.3 -> 21.9 M); not at 4096 or 16384 samples
    if ((dg && dg[0] == '1') || (size_t)M * 8 + fixed_bytes > 80 * 1024 || M == 4096) {
        std::vector<int> fs;
        if (factorize_small(M, &fs) &&
            (size_t)M * 8 + fixed_bytes + (size_t)stage_twiddle_count(M, fs) * 8 <= LDS_BUDGET) {
            *fac = fs;
            *small = true;
            return true;
        }
    }
    return factorize(M, fac);
}

}  // namespace

bool ofx_lds_supported(int n_samples) {
    if (n_samples < 16 || (n_samples % 2)) return false;
    std::vector<int> fac;
    bool small = false;
    if (!choose_factors(n_samples / 2, LDS_FIXED_BYTES, &fac, &small)) return false;
    return lds_bytes_for(n_samples / 2, fac) <= LDS_BUDGET;
}

int ofx_lds_release(ofx_plan* p) {
    if (p->d_lds_tw) (void)hipFree(p->d_lds_tw);
    if (p->d_lds_slots) (void)hipFree(p->d_lds_slots);
    if (p->d_lds_pos) (void)hipFree(p->d_lds_pos);
    if (p->d_lds_spec) (void)hipFree(p->d_lds_spec);
    p->d_lds_spec = nullptr;
    p->lds_spec_bytes = 0;
    p->d_lds_pos = nullptr;
    p->d_lds_tw = nullptr;
    p->d_lds_slots = nullptr;
    return OFX_OK;
}

template <int BT, bool PF, int MAXR = 16>
static int launch_lds(ofx_plan* p, const OfxPlanDev& pd, const LdsGeom& g, int nslots,
                      const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                      hipStream_t st, size_t lds) {
    OFX_LDS_ATTR_ONCE((k_lds<BT, PF, MAXR>), LDS_BUDGET);
    int per_cu = (int)((160 * 1024) / lds);
    const int by_threads = 1024 / BT;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)p->cu_count * per_cu;
    if (nslots > 1) {
        // per-workgroup spectrum scratch (sized for the full grid, allocated once)
        const size_t need = (size_t)grid * g.M * sizeof(float2);
        if (p->lds_spec_bytes < need) {
            if (p->d_lds_spec) (void)hipFree(p->d_lds_spec);
            p->d_lds_spec = nullptr;
            p->lds_spec_bytes = 0;
            OFX_HIP(hipMalloc(&p->d_lds_spec, need));
            p->lds_spec_bytes = need;
        }
    }
    if (grid > n) grid = n;
    size_t tix = 0;
    int rc = ofx_time_begin(p, st, &tix);
    if (rc) return rc;
    hipLaunchKernelGGL((k_lds<BT, PF, MAXR>), dim3((unsigned)grid), dim3(BT), lds, st, pd, g,
                       reinterpret_cast<const LdsSlot*>(p->d_lds_slots), nslots,
                       reinterpret_cast<const float2*>(p->d_lds_tw),
                       reinterpret_cast<const LdsPair*>(p->d_lds_pos), d_traces, d_valid, n, d_out,
                       nslots > 1 ? reinterpret_cast<float2*>(p->d_lds_spec) : nullptr);
    rc = ofx_time_end(p, st, tix);
    if (rc) return rc;
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

int ofx_lds_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n,
                    float* d_out, hipStream_t st) {
    const int N = p->N, M = N / 2;
    std::vector<int> fac;
    bool small_radix = false;
    if (!choose_factors(M, LDS_FIXED_BYTES, &fac, &small_radix) ||
        lds_bytes_for(M, fac) > LDS_BUDGET) {
        ofx_set_error("LDS engine: n_samples=%d is not supported", N);
        return OFX_ERR_UNSUPPORTED;
    }
    OfxPlanDev pd;
    ofx_fill_plan_dev(p, &pd);
    // low bins of V the plan reads back (lowchi2 cut-offs, psd_amp bands): the LDS set aside
    // for them follows the plan, and what it leaves decides how many twiddle tables the
    // radix-8 / 16 stages get (choose_anchor_level)
    int vneed = 0;
    for (int s = 0; s < OFX_MAX_SLOTS; ++s)
        if (p->slot[s].set)
            for (const OfxSearchDev& q : p->slot[s].searches) vneed = std::max(vneed, q.nlow);
    for (const auto& bd : p->bands) vneed = std::max(vneed, bd.k_hi);
    const int vlow_cap = std::min(LDS_VLOW, std::max(64, (vneed + 63) / 64 * 64));
    const size_t other_bytes = (size_t)vlow_cap * 8 + 32 * 4 + 32 * sizeof(OfxCand);
    const int T = choose_anchor_level(M, fac, other_bytes);
    LdsGeom g;
    std::vector<float2> t1;
    const bool rebuild = !p->d_lds_tw || p->lds_tw_level != T;
    lds_layout(M, fac, T, &g, rebuild ? &t1 : nullptr);
    g.vlow = vlow_cap;
    const double c0 = -6.283185307179586476925286766559 / (double)N;
    if (rebuild) {
        if (p->d_lds_tw) {
            OFX_HIP(hipStreamSynchronize(st));     // an earlier launch may still read the tables
            (void)hipFree(p->d_lds_tw);
            p->d_lds_tw = nullptr;
        }
        OFX_HIP(hipMalloc(&p->d_lds_tw, sizeof(float2) * t1.size()));
        OFX_HIP(hipMemcpy(p->d_lds_tw, t1.data(), sizeof(float2) * t1.size(),
                          hipMemcpyHostToDevice));
        p->lds_tw_level = T;
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
    {
        // pair tables: geometry + the slot's filter; rebuilt when the plan's filters changed
        const int np = M / 2 + 1;
        const int ntab = std::max(1, nslots);
        if (p->lds_pair_stamp != p->filter_stamp || p->lds_pair_slots != ntab || !p->d_lds_pos) {
            std::vector<LdsPair> tab((size_t)ntab * np);
            int si = 0;
            for (int s = 0; s < OFX_MAX_SLOTS || si == 0; ++s) {
                const bool have = s < OFX_MAX_SLOTS && p->slot[s].set && !p->slot[s].searches.empty();
                if (s < OFX_MAX_SLOTS && !have && !(nslots == 0 && s == 0)) continue;
                for (int k = 0; k < np; ++k) {
                    const int pidx = (k == 0) ? 0 : M - k;
                    const int kp = (k == 0) ? M : pidx;
                    LdsPair& e = tab[(size_t)si * np + k];
                    memset(&e, 0, sizeof(e));
                    e.k = k;
                    e.pk = pos_of(k, g);
                    e.pp = pos_of(pidx, g);
                    e.tx = (float)std::cos(c0 * k);
                    e.ty = (float)std::sin(c0 * k);
                    if (have) {
                        const OfxSlotHost& h = p->slot[s];
                        e.wkx = (float)h.wf_host[2 * k];
                        e.wky = (float)h.wf_host[2 * k + 1];
                        e.wpx = (float)h.wf_host[2 * kp];
                        e.wpy = (float)h.wf_host[2 * kp + 1];
                        e.gk = (float)(((k == 0) ? 1.0 : 2.0) * h.g_host[k]);
                        e.gp = (kp == k) ? 0.0f : (float)(((kp == M) ? 1.0 : 2.0) * h.g_host[kp]);
                    }
                }
                std::sort(tab.begin() + (size_t)si * np, tab.begin() + (size_t)(si + 1) * np,
                          [](const LdsPair& a, const LdsPair& b) { return a.pk < b.pk; });
                ++si;
                if (si >= ntab) break;
            }
            if (p->d_lds_pos) (void)hipFree(p->d_lds_pos);
            p->d_lds_pos = nullptr;
            OFX_HIP(hipMalloc(&p->d_lds_pos, sizeof(LdsPair) * tab.size()));
            OFX_HIP(hipMemcpy(p->d_lds_pos, tab.data(), sizeof(LdsPair) * tab.size(),
                              hipMemcpyHostToDevice));
            p->lds_pair_stamp = p->filter_stamp;
            p->lds_pair_slots = ntab;
        }
    }
    if (nslots > 0) {
        if (!p->d_lds_slots) OFX_HIP(hipMalloc(&p->d_lds_slots, sizeof(LdsSlot) * OFX_MAX_SLOTS));
        if (p->lds_slot_stamp != p->filter_stamp) {
            OFX_HIP(hipStreamSynchronize(st));  // an earlier launch may still read the table
            const size_t bytes = sizeof(LdsSlot) * (size_t)nslots;
            p->h_slot_args.assign(reinterpret_cast<const unsigned char*>(args.data()),
                                  reinterpret_cast<const unsigned char*>(args.data()) + bytes);
            OFX_HIP(hipMemcpyAsync(p->d_lds_slots, p->h_slot_args.data(), bytes,
                                   hipMemcpyHostToDevice, st));
            p->lds_slot_stamp = p->filter_stamp;
        }
    }
    // about one widest butterfly per thread and stage
    const int bf = M / fac[0];
    const size_t lds = (size_t)M * 8 + other_bytes + (size_t)g.ntw * 8;
    // the register prefetch of the next trace pays when only one workgroup fits a CU
    // (nothing else hides the HBM latency); with many small workgroups it only costs occupancy
    const bool pf = lds > 80 * 1024;
    // one workgroup per CU and no radix-8 / 16 stage (e.g. 25000 samples = 4 * 5^5 points): the
    // small butterflies fit 1024 threads, and sixteen waves hide the LDS latency of the stages
    // (no register prefetch: 99 VGPRs, nothing spilled)
    int max_r = 2;
    for (int r : fac) max_r = std::max(max_r, r);
    if (const char* bt = getenv("OFX_DIAG_LDS_BT")) {
        if (max_r <= 8) {
            const int v = atoi(bt);
            if (v == 1024) return launch_lds<1024, false, 8>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
            if (v == 512) return launch_lds<512, false, 8>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
            if (v == 256) return launch_lds<256, false, 8>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
        }
    }
    if (pf && max_r <= 5)
        return launch_lds<1024, false, 5>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
    if (pf && max_r <= 8)
        return launch_lds<1024, false, 8>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
    if (bf >= 1024 || pf)        // one workgroup per CU: give it eight waves
        return pf ? launch_lds<512, true>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds)
                  : launch_lds<512, false>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
    if (bf >= 256)
        return pf ? launch_lds<256, true>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds)
                  : launch_lds<256, false>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
    if (bf >= 128) return launch_lds<128, false>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
    return launch_lds<64, false>(p, pd, g, nslots, d_traces, d_valid, n, d_out, st, lds);
}

// ------------------------------------------------------------------ standalone batched FFT
struct OfxLdsFft {
    // rows of 16384 / 12500 / 6250 points (32768 / 25000 / 12500-sample traces) go to the
    // register-resident transforms of ofx_fused.hip / ofx_fused25.hip instead (reg: the handle,
    // reg_kind: 32, 25 or 12)
    void* reg = nullptr;
    int reg_kind = 0;
    LdsGeom g;
    float2* d_tw = nullptr;
    int* d_pos = nullptr;
    size_t lds = 0;
    int cu_count = 256;
};

namespace {
template <int BT, bool FWD, int MAXR = 16>
int launch_lds_fft(OfxLdsFft* f, const float2* in, float2* out, long long rows, hipStream_t st) {
    OFX_LDS_ATTR_ONCE((k_lds_fft<BT, FWD, MAXR>), LDS_BUDGET);
    int per_cu = (int)((160 * 1024) / f->lds);
    if (per_cu > 1024 / BT) per_cu = 1024 / BT;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)f->cu_count * per_cu;
    if (grid > rows) grid = rows;
    hipLaunchKernelGGL((k_lds_fft<BT, FWD, MAXR>), dim3((unsigned)grid), dim3(BT), f->lds, st, f->g,
                       f->d_tw, f->d_pos, in, out, rows);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
}  // namespace

int ofx_ldsfft_create(int n_complex, int device, OfxLdsFft** out, bool allow_reg) {
    if (allow_reg) {
        void* h = nullptr;
        int kind = 32;
        int r = ofx_fused_fft_create(n_complex, device, &h);
        if (r == OFX_ERR_UNSUPPORTED) {
            kind = 25;
            r = ofx_fused25_fft_create(n_complex, device, &h);
        }
        if (r == OFX_ERR_UNSUPPORTED) {
            kind = 12;
            r = ofx_fused12_fft_create(n_complex, device, &h);
        }
        if (r == OFX_ERR_UNSUPPORTED) {
            kind = 20;
            r = ofx_fused20_fft_create(n_complex, device, &h);
        }
        if (r == OFX_OK) {
            OfxLdsFft* f = new OfxLdsFft();
            f->reg = h;
            f->reg_kind = kind;
            *out = f;
            return OFX_OK;
        }
        if (r != OFX_ERR_UNSUPPORTED) return r;
    }
    std::vector<int> fac;
    const int M = n_complex;
    bool small = false;
    if (M < 8 || !choose_factors(M, 0, &fac, &small)) return OFX_ERR_UNSUPPORTED;
    const int T = choose_anchor_level(M, fac, 0);
    const size_t lds = (size_t)M * 8 + (size_t)stage_twiddle_count(M, fac, T) * 8;
    if (lds > LDS_BUDGET) return OFX_ERR_UNSUPPORTED;
    OfxLdsFft* f = new OfxLdsFft();
    std::vector<float2> t1;
    lds_layout(M, fac, T, &f->g, &t1);
    f->lds = lds;
    std::vector<int> pos((size_t)M);
    for (int k = 0; k < M; ++k) pos[k] = pos_of(k, f->g);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) f->cu_count = prop.multiProcessorCount;
    if (hipMalloc(&f->d_tw, t1.size() * sizeof(float2)) != hipSuccess ||
        hipMalloc(&f->d_pos, pos.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(f->d_tw, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice) !=
            hipSuccess ||
        hipMemcpy(f->d_pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice) !=
            hipSuccess) {
        ofx_set_error("ofx_ldsfft_create: device allocation failed");
        ofx_ldsfft_destroy(f);
        return OFX_ERR_HIP;
    }
    *out = f;
    return OFX_OK;
}

void ofx_ldsfft_destroy(OfxLdsFft* f) {
    if (!f) return;
    if (f->reg_kind == 32) ofx_fused_fft_destroy(f->reg);
    if (f->reg_kind == 25) ofx_fused25_fft_destroy(f->reg);
    if (f->reg_kind == 12) ofx_fused12_fft_destroy(f->reg);
    if (f->reg_kind == 20) ofx_fused20_fft_destroy(f->reg);
    if (f->d_tw) (void)hipFree(f->d_tw);
    if (f->d_pos) (void)hipFree(f->d_pos);
    delete f;
}

int ofx_ldsfft_exec(OfxLdsFft* f, bool forward, const float2* in, float2* out, long long rows,
                    hipStream_t st) {
    if (rows <= 0) return OFX_OK;
    if (f->reg_kind == 32) return ofx_fused_fft_exec(f->reg, forward, in, out, rows, st);
    if (f->reg_kind == 25) return ofx_fused25_fft_exec(f->reg, forward, in, out, rows, st);
    if (f->reg_kind == 12) return ofx_fused12_fft_exec(f->reg, forward, in, out, rows, st);
    if (f->reg_kind == 20) return ofx_fused20_fft_exec(f->reg, forward, in, out, rows, st);
    const bool big = f->g.M >= 1024;
    int max_r = 2;
    for (int i = 0; i < f->g.nfac; ++i) max_r = std::max(max_r, f->g.fac[i]);
    if (max_r <= 5 && f->g.M >= 8192)         // 16 waves per CU hide the LDS latency of the stages
        return forward ? launch_lds_fft<1024, true, 5>(f, in, out, rows, st)
                       : launch_lds_fft<1024, false, 5>(f, in, out, rows, st);
    if (max_r <= 8 && f->g.M >= 8192)
        return forward ? launch_lds_fft<1024, true, 8>(f, in, out, rows, st)
                       : launch_lds_fft<1024, false, 8>(f, in, out, rows, st);
    if (forward)
        return big ? launch_lds_fft<512, true>(f, in, out, rows, st)
                   : launch_lds_fft<256, true>(f, in, out, rows, st);
    return big ? launch_lds_fft<512, false>(f, in, out, rows, st)
               : launch_lds_fft<256, false>(f, in, out, rows, st);
}
