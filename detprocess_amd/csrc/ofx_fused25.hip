// ofx_fused25.hip -- FUSED engine for the reference example's 25000-sample traces
// (/root/reference/examples/processing/process_example.yaml:93): the register-resident design
// of ofx_fused.hip (load -> real FFT -> optimal filter + chi2_0 -> inverse FFT -> arg-max /
// chi2 / low-frequency chi2 -> one output row, HBM touched once per sample) for a length that
// is not a power of two.  tools/model_fused25.py is the NumPy model of this file's index maps.
//
// Geometry: N = 25000 real samples, M = 12500 packed complex points z[m] = x[2m] + i x[2m+1],
// M = R1 R2 R3 = 20 x 25 x 25:
//   m = 625 n1 + 25 n2 + n3        k = k1 + 20 k2 + 500 k3
//   F1: virtual thread n' = 25 n2 + n3 (625 of them), 20-point DFT over n1 -> k1
//       (prime-factor map, no twiddles), x w_M^{n' k1}
//   E1: LDS exchange D1[k1][n']            (row stride 633 = 25 mod 32: conflict-free)
//   F2: virtual thread (k1, n3) (500), 25-point DFT over n2 -> k2, x w_625^{n3 k2}
//   E2: LDS exchange D2[k1 + 20 k2][n3]    (row stride 25, odd: conflict-free)
//   F3: thread v (250) owns the two 25-point blocks k_low = v and its Hermitian partner
//       500 - v (v = 0: the self-paired blocks 0 and 250): the real-FFT unpack, the filter
//       multiply, chi2_0 and the re-pack need no exchange -- slot J pairs bin v + 500 J with
//       bin M - (v + 500 J), both in this thread.
//   I3 / E3 / I2 / E4 / I1 mirror F3 / E2 / F2 / E1 / F1 with conjugate twiddles.
// After I1 virtual thread n' holds A(n) for the 40 lags n = 1250 n1 + 2 n' + {0,1}.
//
// A workgroup is 256 threads of which 250 work; a thread carries 2 virtual threads in F2
// (50 complex values), one block pair in F3 (50), and 2 or 3 virtual threads in F1 / I1
// (625 = 2.5 x 250: the threads below 125 -- waves 0 and 1 -- take a third one; 60 values).
// Exchanges run in two passes through a half-size buffer (50.6 KB) so that two workgroups are
// resident per CU: E1 / E4 rows k1 < 10 then k1 >= 10 (the F2 threads of round h read exactly
// the rows of pass h, because 250 = 10 x 25), E2 / E3 rows k_low < 250 then >= 250.
// The low-frequency stash holds 1250 bins in LDS (lowchi2_fcutoff up to 62 kHz at 1.25 MHz;
// the example YAML's 50 kHz is 1001 bins); wider plans fall back (OFX_ERR_UNSUPPORTED).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ofx_common.h"
#include "ofx_device.h"
#include "ofx_fft_mixed.h"
#include "ofx_fused_host.h"

using namespace ofxfft;

namespace {

// The file builds for R1 = 20 (25000 samples; comments below quote these numbers) and, included
// from ofx_fused12.hip with OFX25_R1 = 10, for 12500 samples (M = 6250 = 10 x 25 x 25: 125 working
// threads in a 128-thread workgroup, five full rounds of 10-point transforms in F1 / I1 and no
// partial one, four workgroups per CU).
#ifndef OFX25_R1
#define OFX25_R1 20
#endif
#if OFX25_R1 == 20
#define OFX25_FN(x) ofx_fused25_##x
#define k_fused25 k_fused25
#define k_fft25 k_fft25
#elif OFX25_R1 == 16
#define OFX25_FN(x) ofx_fused20_##x
#define k_fused25 k_fused20
#define k_fft25 k_fft20
#else
#define OFX25_FN(x) ofx_fused12_##x
#define k_fused25 k_fused12
#define k_fft25 k_fft12
#endif
constexpr int R1 = OFX25_R1, R2 = 25, R3 = 25;
static_assert(R1 == 20 || R1 == 16 || R1 == 10, "supported first-stage lengths");
constexpr int GM = R1 * R2 * R3;        // 12500 packed complex points
constexpr int GN = 2 * GM;              // 25000 samples
#define GEO_N GN                        // (the names the shared fragments ofx_fused_*.inc use)
#define GEO_WIDE false
#define GEO_LDS_BINS NLOW_MAX
constexpr int GP = R1 * R2;             // 500 blocks of R3 bins
constexpr int GT = GP / 2;              // 250 working threads
constexpr int BLK = (GT + 63) / 64 * 64;    // 256
constexpr int NV1 = R2 * R3;            // 625 virtual threads of F1 / I1
constexpr int NRF = NV1 / GT;           // 2 full rounds of them per thread ...
constexpr int T3 = NV1 - NRF * GT;      // ... and 125 threads carry one more (the "third round")
constexpr bool PART = T3 > 0;
constexpr int NR1 = NRF + (PART ? 1 : 0);
constexpr int W3 = (T3 + 63) / 64;      // waves that hold the partial round
constexpr int NV = NR1 * R1;            // 60 complex registers
static_assert(NV >= 2 * R3, "register array");
constexpr int ROWS = 2 * NV1;           // 1250 samples per register row n1
constexpr int LD1 = 633;                // D1 row stride (= 25 mod 32, >= 625)
constexpr int HR1 = R1 / 2;             // D1 rows per pass
static_assert(GT == HR1 * R3 && R1 * R3 == 2 * GT, "the F2 threads of round h read the rows of pass h");
constexpr int XB_ELEMS = HR1 * LD1;     // 6330 complex >= 250 * 25
constexpr int NLOW_MAX = 5 * GP / 2;    // 1250 low bins 2 X_k kept in LDS
constexpr int VPAD = 256;               // table row length (threads)
constexpr int T1PAD = 640;              // anchor table row length (virtual threads of F1)
constexpr int NWAVE = BLK / OFX_WAVE;
constexpr int WG_PER_CU = 512 / BLK;    // 2 (256 VGPRs per thread)
// The third round of I1 / the tail (virtual threads 500..624) sits in the waves from I3W on:
// 0 = waves 0, 1 as in F1; 2 = waves 2, 3, so that over a whole trace every wave -- every SIMD,
// which hosts the same wave of both resident workgroups -- carries five rounds of 20-point
// transforms instead of six on SIMDs 0, 1 and four on SIMDs 2, 3.
#ifndef OFX25_I3W
#define OFX25_I3W 0
#endif
constexpr int I3W = OFX25_I3W;
constexpr int I3T = PART ? 64 * I3W : 0;    // first thread of the inverse third round
static_assert(XB_ELEMS >= GT * R3 && XB_ELEMS * 2 >= GM, "exchange buffer");

#ifndef OFX_XPRIO
#define OFX_XPRIO 2
#endif
#ifndef OFX_TPRIO
#define OFX_TPRIO 1
#endif
#ifndef OFX_MPRIO
#define OFX_MPRIO 1
#endif

struct Shared25 {
    float xb[2 * XB_ELEMS];        // exchange buffer (complex) / lag dump (real, 12500 per pass)
    cpx t2[R2 * R3];               // w_625^{n3 k2}, index k2 * 25 + n3
};
struct Lds25 {
    cpx xlow[NLOW_MAX + 6];        // 2 X_k for k < 1250
    float red[4][NWAVE];
    float tdred[2][OFX_MAX_TDWIN][4][NWAVE];   // double-buffered by trace parity (see ofx_fused.hip)
    OfxCand cand[NWAVE];
    OfxCand wc[OFX_MAX_SEARCHES][NWAVE];
    OfxCand fin[OFX_MAX_SEARCHES];
    float lowp[OFX_MAX_SEARCHES][NWAVE];
    float bcast[8];
    cpx perm[2 * R3];
    float nb[OFX_MAX_SEARCHES][2];
    OfxRefined ref[OFX_MAX_SEARCHES];
};
constexpr size_t LDS_BYTES = sizeof(Shared25) + sizeof(Lds25);
static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS budget");

#include "ofx_fused_stamps.h"

struct Tabs25 {
    const float2* t1;     // [2][640] float4 rows of stage-1 twiddle anchors
    const float2* t2;     // [25][25]   w_625^{n3 k2}
    const float4* midW;   // [25][256]  (W_k / 2, conj(W_p) / 2)   slot J, thread v
    const float2* midG;   // [25][256]  (g_k', g_p')
    const float2* tbase;  // [256]      T_v = i exp(-2 pi i v / N); T of slot J is T_v w_50^J
    // Base of thread 0 for its slots J >= 13 (block 250); a constant of the geometry.  The kernel reads
    // it from its ARGUMENT `tabs`, never from the per-slot copy in global memory: with
    // `(tl == 0) ? slots[i].tabs.tb0hi : tb` the compiler sank the scalar load into an if / else, and
    // in the flow block of that if / else the register allocator put live-range-split copies ABOVE
    // the instruction that restores EXEC (behind a rematerialised s_movk_i32, where
    // SIInstrInfo::isBasicBlockPrologue stops looking), so thread 0 alone kept stale values in some
    // of its 50 registers: round 2's wrong <2, true> / <6, true> (DESIGN.md section 5.1b;
    // tools/isa_hazards.py `execprologue` finds the pattern in the ISA, `make check`).
    float2 tb0hi;
    unsigned rowmask;     // register rows n1 (1250 lags each) the slot's windowed searches touch
    float2 wq;            // W_{M/2}
    float gq;             // g_{M/2}
};
struct SlotArg25 {
    OfxSlotDev sd;
    Tabs25 tabs;
};

#include "ofx_fused_parts.h"
// (plain fmaxf / fminf: the compiler forms v_max3_f32 / v_min3_f32 itself)
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(a, fminf(b, c)); }

// Thread 0 owns the self-paired blocks A0 = block 0 (bins 500 j) and B0 = block 250
// (bins 250 + 500 j).  A permutation of its 50 values brings them to the generic slot shape
// "slot J pairs (d[J], d[25 + 24 - J])":
//   genA = [A0[0..12], B0[0..11]]      genB = [B0[13..24], A0[13..24], A0[0]]
// (slot 0: DC with Nyquist; slots 1..12: A0[J] with A0[25 - J]; slots 13..24: B0[J - 13] with
// B0[24 - (J - 13)]); B0[12] is the self-paired bin k = M/2, handled by the caller.
__device__ constexpr int perm_in_src(int j) {
    if (j < 13) return j;
    if (j < 25) return j + 12;
    if (j < 37) return j + 13;
    if (j < 49) return j - 24;
    return 0;
}
__device__ constexpr int perm_out_src(int j) {
    if (j < 13) return j;
    if (j < 25) return 24 + j;
    if (j < 37) return j - 12;
    return j - 13;          // j = 37 is overwritten by the caller
}
__device__ __forceinline__ void perm_in(cpx (&d)[NV], bool z, cpx* buf) {
    if (z) {
#pragma unroll
        for (int j = 0; j < 50; ++j) buf[j] = d[j];
#pragma unroll
        for (int j = 13; j < 50; ++j) d[j] = buf[perm_in_src(j)];
    }
}
__device__ __forceinline__ cpx perm_out(cpx (&d)[NV], bool z, cpx aself, const Tabs25& tabs,
                                        cpx chi, cpx* buf) {
    if (z) {
        const cpx zq = cmulc(aself, mk(tabs.wq.x, tabs.wq.y));
        chi = pfma(aself * aself, mk(2.0f * tabs.gq, 2.0f * tabs.gq), chi);
#pragma unroll
        for (int j = 0; j < 50; ++j) buf[j] = d[j];
#pragma unroll
        for (int j = 13; j < 50; ++j) d[j] = buf[perm_out_src(j)];
        d[37] = zq + zq;
    }
    return chi;
}

#ifndef OFX_MID_DEPTH
#define OFX_MID_DEPTH 4
#endif
constexpr int MID_DEPTH = OFX_MID_DEPTH;
struct MidRsrc {
    __amdgpu_buffer_rsrc_t w, g;
};
// 25 pair slots; table rows software-pipelined MID_DEPTH slots ahead.  Who holds which low
// bin (stashed as 2 X_k in L.xlow for the low-frequency chi2 and psd_amp): slot J of thread v
// has xk2 = 2 X_k, k = v + 500 J, and xp2 = 2 conj(X_p), p = 500 (25 - J) - v (v != 0);
// thread 0's slots 13 and 14 hold the bins 250 and 750.
template <int J, int NB>
__device__ __forceinline__ void mid_unrolled(cpx (&d)[NV], const MidRsrc& r, int v, Lds25& L, cpx tlo, cpx thi, float4 (&tw)[NB],
                                             cpx (&tg)[NB], cpx& chi) {
    if constexpr (J < R3) {
        if constexpr (J + MID_DEPTH < R3) {
            tw[(J + MID_DEPTH) % NB] = buf_ld4(r.w, v * 16, (J + MID_DEPTH) * VPAD * 16);
            tg[(J + MID_DEPTH) % NB] = buf_ld2(r.g, v * 8, (J + MID_DEPTH) * VPAD * 8);
        }
        const cpx T = twmul50<J, -1>(J < 13 ? tlo : thi);
        cpx xk2, xp2;
        mid_slot(d[J], d[R3 + R3 - 1 - J], T, tw[J % NB], tg[J % NB], xk2, xp2, chi);
        // (no branches: idle lanes mirror lane 249, and the entries that do not apply to a
        // thread go to the padding behind the stash)
        if constexpr (J <= 2) L.xlow[v + GP * J] = xk2;
        if constexpr (J >= 23) L.xlow[v != 0 ? GP * (R3 - J) - v : NLOW_MAX + 1] = cconj(xp2);
        if constexpr (J == 13 || J == 14)
            L.xlow[v == 0 ? GP / 2 + GP * (J - 13) : NLOW_MAX + 2] = xk2;
        mid_unrolled<J + 1, NB>(d, r, v, L, tlo, thi, tw, tg, chi);
    }
}
__device__ __forceinline__ cpx middle_slots(cpx (&d)[NV], const MidRsrc& r, int v, Lds25& L, cpx tlo, cpx thi, cpx chi) {
    constexpr int NB = MID_DEPTH + 1;
    float4 tw[NB];
    cpx tg[NB];
#pragma unroll
    for (int j = 0; j < MID_DEPTH; ++j) {
        tw[j] = buf_ld4(r.w, v * 16, j * VPAD * 16);
        tg[j] = buf_ld2(r.g, v * 8, j * VPAD * 8);
    }
    mid_unrolled<0, NB>(d, r, v, L, tlo, thi, tw, tg, chi);
    return chi;
}

// Inter-stage twiddles w_M^{n' k1}, k1 = 5 a + b = 0..19, from four anchors per virtual thread:
// B1 = w^{n'}, B2 = w^{2 n'}, A1 = w^{5 n'}, A2 = w^{10 n'} (32 bytes, one L2 round trip);
// B3 = B1 B2, B4 = B2 B2, A3 = A1 A2, w^{n' (5 a + b)} = A_a B_b: at most three roundings on top
// of the table's.  (Seven tabulated anchors cost 16 registers per round of virtual threads, and
// with three rounds in flight around E4 the third set was spilled.)  t1a[0][n'] = (B1, B2),
// t1a[1][n'] = (A1, A2).
struct T1Anch {
    float4 q[2];
};
__device__ __forceinline__ T1Anch t1_load(__amdgpu_buffer_rsrc_t t1a, int vt) {
    T1Anch r;
#pragma unroll
    for (int c = 0; c < 2; ++c) r.q[c] = buf_ld4(t1a, vt * 16, c * T1PAD * 16);
    return r;
}
template <bool CONJ, int O>
__device__ __forceinline__ void t1_apply(cpx (&d)[NV], const T1Anch& an) {
#ifdef ABL_NOFFT
    d[O] = d[O] + lo2(an.q[0]) + lo2(an.q[1]);
    return;
#endif
    cpx B[5], A[4];
    B[1] = lo2(an.q[0]);
    B[2] = hi2(an.q[0]);
    B[3] = cmul(B[1], B[2]);
    B[4] = cmul(B[2], B[2]);
    A[1] = lo2(an.q[1]);
    A[2] = hi2(an.q[1]);
    if constexpr (R1 > 15) A[3] = cmul(A[1], A[2]);
#pragma unroll
    for (int a = 0; a < (R1 + 4) / 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int k1 = 5 * a + b;
            if (k1 == 0 || k1 >= R1) continue;
            const cpx w = (a == 0) ? B[b] : (b == 0) ? A[a] : cmul(A[a], B[b]);
            d[O + k1] = CONJ ? cmulc(d[O + k1], w) : cmul(d[O + k1], w);
        }
}

// Low-frequency chi2 (ofx_device.h: ofx_lowchi2_term) with the trace length as a compile-time
// constant.  The bins of a thread are GT apart, so the phase exp(-2 pi i k (d + frac) / N) of a search
// runs along a chain: lowchi2_phase(tt) and the uniform step lowchi2_phase(GT) from sincospif (integer
// part of the angle reduced exactly: k d < 2^31 for k < 1250, the remainder by GN a multiply-high),
// every further bin one complex product -- two sincospif per search and thread instead of five.
__device__ __forceinline__ cpx lowchi2_phase(int k, int dl, float frac) {
    int m = (k * dl) % GN;
    m = m < 0 ? m + GN : m;
    float sn, cs;
    sincospif(-2.0f * ((float)m + (float)k * frac) / (float)GN, &sn, &cs);
    return mk(cs, sn);
}
__device__ __forceinline__ float lowchi2_term(int k, cpx ph, float amp, cpx x2, cpx S, float g) {
    const float pr = ph.x * S.x - ph.y * S.y;
    const float pi = ph.x * S.y + ph.y * S.x;
    const float rr = 0.5f * x2.x - amp * pr;
    const float ri = 0.5f * x2.y - amp * pi;
    const float w = (k == 0) ? 1.0f : 2.0f;
    return w * g * (rr * rr + ri * ri);
}

template <int DIR, int OFF>
__device__ __forceinline__ void dft_r1(cpx (&d)[NV]) {
    if constexpr (R1 == 20) dft20<DIR, NV, OFF>(d);
    else if constexpr (R1 == 16) dft<16, DIR, NV, OFF>(d);
    else dft10<DIR, NV, OFF>(d);
}
// the full rounds of F1 / I1: first-stage transforms and inter-stage twiddles of rounds H .. NRF-1
template <int DIR, int H = 0>
__device__ __forceinline__ void r1_dfts(cpx (&d)[NV]) {
    if constexpr (H < NRF) {
        dft_r1<DIR, R1 * H>(d);
        r1_dfts<DIR, H + 1>(d);
    }
}
template <bool CONJ, int H = 0>
__device__ __forceinline__ void r1_twiddles(cpx (&d)[NV], const T1Anch (&g)[NRF]) {
    if constexpr (H < NRF) {
        t1_apply<CONJ, R1 * H>(d, g[H]);
        r1_twiddles<CONJ, H + 1>(d, g);
    }
}

// ------------------------------------------------------------------ the kernel
// FEAT bit 0: searches that are not full-range (scan of the LDS lag dump) or interpolate
// FEAT bit 1: time-domain windows          FEAT bit 2: channel algebra on load
// MULTI: several filter slots share the forward transform (spectrum parked per workgroup)
template <int FEAT, bool MULTI>
__global__ __launch_bounds__(BLK, WG_PER_CU * NWAVE / 4) void k_fused25(   // (threads, waves per SIMD)
    OfxPlanDev pd, OfxSlotDev sd, Tabs25 tabs, const float* __restrict__ traces,
    const uint8_t* __restrict__ valid, long long n_traces, float* __restrict__ out,
    const SlotArg25* __restrict__ slots, int nslots, float2* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Shared25& SH = *reinterpret_cast<Shared25*>(smem_raw);
    Lds25& L = *reinterpret_cast<Lds25*>(smem_raw + sizeof(Shared25));
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool w2 = PART && wave < W3;              // waves holding a third F1 virtual thread
    const bool w2i = PART && ((I3W == 0) ? w2 : !w2);   // ... and a third I1 virtual thread
    const int pre = pd.pre;
    const __amdgpu_buffer_rsrc_t t1q = make_rsrc(tabs.t1, 2 * T1PAD * 16);
    const __amdgpu_buffer_rsrc_t rtb = make_rsrc(tabs.tbase, VPAD * 8);

    for (int i = tid; i < R2 * R3; i += BLK) SH.t2[i] = mk(tabs.t2[i].x, tabs.t2[i].y);

    const size_t ev_stride = (size_t)pd.n_channels * GN;
    cpx d[NV];
    cpx* const xc = reinterpret_cast<cpx*>(SH.xb);

    // Trace load: F1 virtual thread vt reads z[625 n1 + vt]; the third round only in waves 0, 1.
    // Idle lanes (tid >= 250; third round: tid >= 125) are exact mirrors of lane 249 (of virtual
    // thread 624): they load, compute and store the same values to the same places, so that no
    // store needs a guard (a guard is a branch, and a branch splits the scheduling region:
    // spills); only the sums of the reductions mask them.
    auto load_rows = [&](const __amdgpu_buffer_rsrc_t rz, int tl) __attribute__((always_inline)) {
        const int tcl = min(tl, GT - 1);
        const int v3l = min(tl + NRF * GT, NV1 - 1);
#pragma unroll
        for (int h = 0; h < NRF; ++h)
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1)
                d[R1 * h + n1] = buf_ld2(rz, (tcl + GT * h) * 8, n1 * NV1 * 8);
        if constexpr (PART) {
            // (plans with time-domain windows: the waves without a third round load a copy of
            // virtual thread 624 -- mirrors, like the idle lanes -- so that the window sums below
            // need neither a branch nor a mask on these registers)
            if (w2 || (FEAT & 2)) {
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1)
                    d[NRF * R1 + n1] = buf_ld2(rz, v3l * 8, n1 * NV1 * 8);
            } else {   // (defined on every path: otherwise the registers stay live around the loop)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) d[NRF * R1 + n1] = mk(0.0f, 0.0f);
            }
        }
    };
    auto load_trace = [&](long long bb) __attribute__((always_inline)) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const float* e = traces + (size_t)bb * ev_stride;
        load_rows(make_rsrc(e + ((FEAT & 4) ? (size_t)pd.chan[0] * GN : 0), GN * 4), tl);
    };
    auto combine_terms = [&](long long bb) __attribute__((always_inline)) {
        if constexpr (FEAT & 4) {
            if (pd.n_terms == 1 && pd.weight[0] == 1.0f) return;
            int tl = tid;
            asm volatile("" : "+v"(tl));
            const int tcl = min(tl, GT - 1);
            const int v3l = min(tl + NRF * GT, NV1 - 1);
            const float* e = traces + (size_t)bb * ev_stride;
            const float w0 = pd.weight[0];
#pragma unroll
            for (int j = 0; j < NV; ++j) d[j] = d[j] * mk(w0, w0);
            for (int c = 1; c < pd.n_terms; ++c) {
                const __amdgpu_buffer_rsrc_t rz = make_rsrc(e + (size_t)pd.chan[c] * GN, GN * 4);
                const float wgt = pd.weight[c];
#pragma unroll
                for (int h = 0; h < NRF; ++h)
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) {
                        const cpx s = buf_ld2(rz, (tcl + GT * h) * 8, n1 * NV1 * 8);
                        d[R1 * h + n1] = pfma(mk(wgt, wgt), s, d[R1 * h + n1]);
                    }
                if (PART && (w2 || (FEAT & 2))) {
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) {
                        const cpx s = buf_ld2(rz, v3l * 8, n1 * NV1 * 8);
                        d[(NR1 - 1) * R1 + n1] = pfma(mk(wgt, wgt), s, d[(NR1 - 1) * R1 + n1]);
                    }
                }
            }
        }
    };

    // MULTI: the spectrum of the current trace, [value j][thread] in this workgroup's area.
    const __amdgpu_buffer_rsrc_t rspec =
        make_rsrc(spec + (MULTI ? (size_t)blockIdx.x * 2 * R3 * BLK : 0), 2 * R3 * BLK * 8);
    auto store_spec = [&]() __attribute__((always_inline)) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int j = 0; j < 2 * R3; ++j) {
            u32x2 v;
            v.x = __float_as_uint(d[j].x);
            v.y = __float_as_uint(d[j].y);
            __builtin_amdgcn_raw_buffer_store_b64(v, rspec, tl * 8, j * BLK * 8, 0);
        }
    };
    auto load_spec = [&]() __attribute__((always_inline)) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int j = 0; j < 2 * R3; ++j) d[j] = buf_ld2(rspec, tl * 8, j * BLK * 8);
    };

    // pass of D2 row k1u + 20 k2 (k1u = k1l + 10 h): rows < 250 -> 0.  Compile-time in (h, k2).
    auto pass2 = [](int h, int k2) { return k2 < 12 ? 0 : (k2 > 12 ? 1 : h); };

    bool have = false;
    [[maybe_unused]] int tdpar = 0;          // parity of the trace count: which half of L.tdred is written
#ifdef OFX_STAMPS
    int stamp_it = 0;
    unsigned long long* stamp_base;
    {
        const size_t off = (((size_t)blockIdx.x * OFX_STAMP_TRACES) * NWAVE + (size_t)wave) * 16;
        const unsigned long long a = reinterpret_cast<unsigned long long>(spec) + off * 8;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        stamp_base = reinterpret_cast<unsigned long long*>(((unsigned long long)hi << 32) | lo);
    }
#endif
    const long long stride = (long long)gridDim.x;
    for (long long b = (long long)blockIdx.x; b < n_traces; b += stride) {
        float* row = out + (size_t)b * pd.row;
        const bool skip = (valid && !valid[b]);
        if (skip) {
            for (int j = tid; j < pd.row; j += BLK) row[j] = OFX_SENTINEL;
            have = false;
            continue;
        }
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const bool act = tl < GT;                  // this lane works (250 of 256)
        const bool act3 = tl < T3;                 // ... and carries a third F1 virtual thread
        const int tc = min(tl, GT - 1);            // role index (idle lanes mirror lane 249)

        const int vt3 = min(tl + NRF * GT, NV1 - 1);    // third F1 virtual thread (idle: mirror of 624)
        STAMP(0);
        if (!have) load_trace(b);
        combine_terms(b);

        // ------------------------------------------------ time-domain windows
        // Sample index of d[20 h + n1].{x,y} is 1250 n1 + 2 vt + {0,1}, vt = tid + 250 h.
        if constexpr (FEAT & 2) {
            tdpar ^= 1;
#include "ofx_fused_td_endpoints.inc"
            for (int w = 0; w < pd.n_tdwin; ++w) {
                const int lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
                // sums of rounds 0, 1 (sa, sqa: masked by `act` at the end -- idle lanes are mirrors)
                // and of round 2 (sb, sqb: masked by `act3`); duplicates do not change max / min
                float sa = 0.0f, sqa = 0.0f, sb = 0.0f, sqb = 0.0f, mx = -INFINITY, mn = INFINITY;
                cpx s2a = mk(0.0f, 0.0f), sq2a = mk(0.0f, 0.0f), s2b = mk(0.0f, 0.0f), sq2b = mk(0.0f, 0.0f);
                const unsigned fullm = pd.tdw[w].full, anym = fullm | pd.tdw[w].edge;
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) {
                    const int r0 = ROWS * n1;
                    if (!((anym >> n1) & 1u)) continue;                   // uniform: outside
                    if ((fullm >> n1) & 1u) {                             // uniform: full row
#pragma unroll
                        for (int h = 0; h < NRF; ++h) {
                            // (dependent forms only, as in ofx_fused.hip)
                            const cpx v = d[R1 * h + n1];
                            s2a = s2a + v;
                            sq2a = pfma(v, v, sq2a);
                            mx = max3f(mx, v.x, v.y);
                            mn = min3f(mn, v.x, v.y);
                        }
                        if constexpr (PART) {
                            // (every wave, no branch: the waves without a third round hold a copy of
                            // virtual thread 624 there, see load_rows; sums are masked by `act3` at
                            // the end, copies cannot change max / min.  Measured against the form with
                            // a wave-uniform branch around these lines, round 3, same box: 13.5
                            // against 12.1 M traces/s with five windows.)
                            const cpx v = d[NRF * R1 + n1];
                            s2b = s2b + v;
                            sq2b = pfma(v, v, sq2b);
                            mx = max3f(mx, v.x, v.y);
                            mn = min3f(mn, v.x, v.y);
                        }
                    } else {                                              // edge row
                        // (offsets in the row against scalar bounds: see the same lines of ofx_fused.hip)
                        const int lo_r = lo - r0, hi_r = hi - r0;         // uniform
#pragma unroll
                        for (int h = 0; h < NR1; ++h) {
                            const int c = 2 * (h == NRF ? vt3 : tc + GT * h);
                            const bool in0 = (c >= lo_r) && (c < hi_r);
                            const bool in1 = (c >= lo_r - 1) && (c < hi_r - 1);
                            const cpx v = d[R1 * h + n1];
                            const float y0 = in0 ? v.x : 0.0f, y1 = in1 ? v.y : 0.0f;
                            if (h == NRF) {
                                sb = (sb + y0) + y1;
                                sqb = fmaf(y0, y0, fmaf(y1, y1, sqb));
                            } else {
                                sa = (sa + y0) + y1;
                                sqa = fmaf(y0, y0, fmaf(y1, y1, sqa));
                            }
                            mx = max3f(mx, in0 ? v.x : -INFINITY, in1 ? v.y : -INFINITY);
                            mn = min3f(mn, in0 ? v.x : INFINITY, in1 ? v.y : INFINITY);
                        }
                    }
                }
                float s = (act ? sa + (s2a.x + s2a.y) : 0.0f) + (act3 ? sb + (s2b.x + s2b.y) : 0.0f);
                float sq = (act ? sqa + (sq2a.x + sq2a.y) : 0.0f) + (act3 ? sqb + (sq2b.x + sq2b.y) : 0.0f);
                s = ofx_wave_sum(s);
                sq = ofx_wave_sum(sq);
                mx = ofx_wave_max(mx);
                mn = ofx_wave_min(mn);
                if ((tl & 63) == 0) {
                    L.tdred[tdpar][w][0][wave] = s;
                    L.tdred[tdpar][w][1][wave] = mx;
                    L.tdred[tdpar][w][2][wave] = mn;
                    L.tdred[tdpar][w][3][wave] = sq;
                }
            }
            __syncthreads();
#include "ofx_fused_td_finalize.inc"
        }
        if ((MULTI ? nslots : sd.n_search) == 0) {
            have = false;
            continue;
        }
#ifdef ABL_LOADONLY
        {
            cpx acc = mk(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NRF * R1; ++j) acc += d[j];
            if (w2) {
#pragma unroll
                for (int j = NRF * R1; j < NV; ++j) acc += d[j];
            }
            if (acc.x + acc.y == 1.2345f) row[0] = acc.x;
            have = false;
            continue;
        }
#endif
        STAMP(1);
        // ---------------------------------------------------------------- F1
        {
            T1Anch gf[NRF];
#pragma unroll
            for (int h = 0; h < NRF; ++h) gf[h] = t1_load(t1q, tc + GT * h);
            __builtin_amdgcn_sched_barrier(0);
            r1_dfts<-1>(d);
            r1_twiddles<false>(d, gf);
            if (w2) {
                const T1Anch g2 = t1_load(t1q, vt3);
                dft_r1<-1, (NR1 - 1) * R1>(d);
                t1_apply<false, (NR1 - 1) * R1>(d, g2);
            }
        }
        STAMP(2);
        // F2 role of this thread: vt2 = tid + 250 h = 25 k1u + n3u, k1u = k1l + 10 h
        const int k1l = (tc * 1311) >> 15;                  // tc / 25 for tc < 2048
        const int n3u = tc - 25 * k1l;
        const int rb1 = k1l * LD1 + n3u;                    // D1 read / write base (pass = h)
        const int rbB = (tc == 0) ? 0 : (GT - tc);          // partner block row within pass 1
        cpx nd[NV];
        // ---------------------------------------------------------------- E1
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(OFX_XPRIO);
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < NRF; ++h)
#pragma unroll
                for (int j = 0; j < HR1; ++j) xc[j * LD1 + tc + GT * h] = d[R1 * h + HR1 * p + j];
            if (w2) {
#pragma unroll
                for (int j = 0; j < HR1; ++j) xc[j * LD1 + vt3] = d[(NR1 - 1) * R1 + HR1 * p + j];
            }
            __syncthreads();
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) nd[R2 * p + n2] = xc[rb1 + R3 * n2];
        }
#pragma unroll
        for (int j = 0; j < 2 * R2; ++j) d[j] = nd[j];
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(3);
        // ---------------------------------------------------------------- F2
        dft25<-1, NV, 0>(d);
        dft25<-1, NV, R2>(d);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k2 = 1; k2 < R2; ++k2)
                d[R2 * h + k2] = cmul(d[R2 * h + k2], SH.t2[k2 * R3 + n3u]);
        STAMP(4);
        // ---------------------------------------------------------------- E2
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(OFX_XPRIO);
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < R2; ++k2)
                    if (pass2(h, k2) == p)
                        xc[(k1l + HR1 * h + R1 * k2 - GT * p) * R3 + n3u] = d[R2 * h + k2];
            __syncthreads();
            const int rr = (p == 0) ? tc : rbB;
#pragma unroll
            for (int j = 0; j < R3; ++j) nd[R3 * p + j] = xc[rr * R3 + j];
        }
#pragma unroll
        for (int j = 0; j < 2 * R3; ++j) d[j] = nd[j];
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(5);
#if OFX_MPRIO
        __builtin_amdgcn_s_setprio(OFX_MPRIO);
#endif
#define SDX (MULTI ? slots[slot_i].sd : sd)
#define TBX (MULTI ? slots[slot_i].tabs : tabs)
        if constexpr (MULTI) {
            dft25<-1, NV, 0>(d);
            dft25<-1, NV, R3>(d);
            store_spec();
        }
        const int slot_n = MULTI ? nslots : 1;
        for (int slot_i = 0; slot_i < slot_n; ++slot_i) {
        // ------------------------------------------- F3, middle, I3 (registers)
        const MidRsrc rmid = {make_rsrc(TBX.midW, R3 * VPAD * 16), make_rsrc(TBX.midG, R3 * VPAD * 8)};
        cpx chi2v = mk(0.0f, 0.0f);
        if constexpr (!MULTI) {
            dft25<-1, NV, 0>(d);
            dft25<-1, NV, R3>(d);
        }
        {
            const cpx aself = d[R3 + 12];
            if (wave == 0) perm_in(d, tl == 0, L.perm);
            const cpx tb = buf_ld2(rtb, tc * 8, 0);
            // (tabs.tb0hi is a kernel argument -- the same for every slot -- so that this is a
            // select on two registers, not an if / else around a load: see the note at Tabs25::tb0hi)
            const cpx tbh = (tl == 0) ? mk(tabs.tb0hi.x, tabs.tb0hi.y) : tb;
            chi2v = middle_slots(d, rmid, tc, L, tb, tbh, chi2v);
            if (wave == 0) chi2v = perm_out(d, tl == 0, aself, TBX, chi2v, L.perm);
        }
        dft25<+1, NV, 0>(d);
        dft25<+1, NV, R3>(d);
        const float chi0p = act ? chi2v.x + chi2v.y : 0.0f;
#if OFX_MPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(6);
        int tl2 = tid;
        asm volatile("" : "+v"(tl2));
        const int tc2 = min(tl2, GT - 1);
        const int k1l2 = (tc2 * 1311) >> 15;
        const int n3u2 = tc2 - 25 * k1l2;
        const int rb1b = k1l2 * LD1 + n3u2;
        const int rbB2 = (tc2 == 0) ? 0 : (GT - tc2);
        // ---------------------------------------------------------------- E3
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(OFX_XPRIO);
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();
            {
                const int rr = (p == 0) ? tc2 : rbB2;
#pragma unroll
                for (int j = 0; j < R3; ++j) xc[rr * R3 + j] = d[R3 * p + j];
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < R2; ++k2)
                    if (pass2(h, k2) == p)
                        nd[R2 * h + k2] = xc[(k1l2 + HR1 * h + R1 * k2 - GT * p) * R3 + n3u2];
        }
#pragma unroll
        for (int j = 0; j < 2 * R2; ++j) d[j] = nd[j];
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(7);
        // ---------------------------------------------------------------- I2
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k2 = 1; k2 < R2; ++k2)
                d[R2 * h + k2] = cmulc(d[R2 * h + k2], SH.t2[k2 * R3 + n3u2]);
        dft25<+1, NV, 0>(d);
        dft25<+1, NV, R2>(d);
        STAMP(8);
        {
            const int vt3b = min(max(tl2 - I3T, 0) + NRF * GT, NV1 - 1);
            // anchors of the stage-1 twiddles: requested ahead of the exchange (L2 latency)
            T1Anch gi[NRF];
#pragma unroll
            for (int h = 0; h < NRF; ++h) gi[h] = t1_load(t1q, tc2 + GT * h);
            __builtin_amdgcn_sched_barrier(0);
            // ------------------------------------------------------------ E4
#if OFX_XPRIO
            __builtin_amdgcn_s_setprio(OFX_XPRIO);
#endif
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                __syncthreads();
#pragma unroll
                for (int n2 = 0; n2 < R2; ++n2) xc[rb1b + R3 * n2] = d[R2 * p + n2];
                __syncthreads();
#pragma unroll
                for (int h = 0; h < NRF; ++h)
#pragma unroll
                    for (int j = 0; j < HR1; ++j)
                        nd[R1 * h + HR1 * p + j] = xc[j * LD1 + tc2 + GT * h];
                // (third round: the rows of pass 1 stay in the buffer until the next barrier and
                // are read when their turn comes in I1 -- ten values less to hold meanwhile)
                if (p == 0 && w2i) {
#pragma unroll
                    for (int j = 0; j < HR1; ++j) nd[(NR1 - 1) * R1 + j] = xc[j * LD1 + vt3b];
                }
            }
#pragma unroll
            for (int j = 0; j < NRF * R1; ++j) d[j] = nd[j];
#if OFX_XPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            STAMP(9);
            // ------------------------------------------------------------ I1
            r1_twiddles<true>(d, gi);
            // (the third round's anchors: requested once those of the first two are dead, ahead
            // of the two transforms)
            T1Anch g2 = gi[NRF - 1];  // (defined on every path: an undefined value is carried around the loop)
            if (w2i) g2 = t1_load(t1q, vt3b);
            __builtin_amdgcn_sched_barrier(0);
            r1_dfts<+1>(d);
            if constexpr (PART) {
                if (w2i) {
#pragma unroll
                    for (int j = 0; j < HR1; ++j) {
                        d[NRF * R1 + j] = nd[NRF * R1 + j];
                        d[NRF * R1 + HR1 + j] = xc[j * LD1 + vt3b];
                    }
                    t1_apply<true, NRF * R1>(d, g2);
                    dft_r1<+1, NRF * R1>(d);
                } else {
#pragma unroll
                    for (int j = NRF * R1; j < NV; ++j) d[j] = mk(0.0f, 0.0f);
                }
            }
        }
        // d[20 h + n1] = (A(1250 n1 + 2 vt), A(1250 n1 + 2 vt + 1)),  vt = tid + 250 h (idle lanes:
        // copies of the lags of virtual threads 249, 499 and 624; waves 2, 3: zeros in round 3)

#ifdef ABL_NOTAIL
        {
            float acc = chi0p;
#pragma unroll
            for (int j = 0; j < NV; ++j) acc += d[j].x + d[j].y;
            if (acc == 1.2345f) row[0] = acc;
            continue;
        }
#endif
        STAMP(10);
#if OFX_TPRIO
        __builtin_amdgcn_s_setprio(OFX_TPRIO);
#endif
        // ------------------------------------------------------------- tail
        int tt = tid;
        asm volatile("" : "+v"(tt));
        const int lane_t = tt & 63, wave_t = tt >> 6;
        const bool act_t = tt < GT;
        const int tct = min(tt, GT - 1);
        const int vt3t = min(max(tt - I3T, 0) + NRF * GT, NV1 - 1);
        const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(SDX.s, NLOW_MAX * 8);
        const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(SDX.g, NLOW_MAX * 4);
        constexpr int GS = (R1 == 10) ? 10 : R1 / 2;   // registers per group (two groups per round)
        constexpr int NG = NV / GS;
        float gm[NG];
        float mloc = 0.0f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float m = 0.0f;
#pragma unroll
            for (int j = GS * g; j < GS * g + GS; ++j) {
                const cpx sq = d[j] * d[j];
                m = fmaxf(m, fmaxf(sq.x, sq.y));
            }
            gm[g] = m;
            mloc = fmaxf(mloc, m);
        }
        // low-frequency chi2 tables for this thread's bins: requested now, used at the end
        constexpr int NLK = NLOW_MAX / GT;         // 5 low bins per thread
        cpx lk_s[NLK];
        float lk_g[NLK];
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            const int k = (act_t ? tt : 0) + GT * i;
            lk_s[i] = buf_ld2(rs_s, k * 8, 0);
            lk_g[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_g, k * 4, 0, 0));
        }
        // One reduction for the maximum, its rolled index and chi2_0: every wave resolves the
        // smallest rolled index among ITS lags of maximal A^2 (the lanes that hold the wave's
        // maximum search only the matching group of registers), the four candidates meet in LDS
        // behind a single barrier.  (L.cand / L.red were last read in the previous tail, at least
        // four exchange barriers ago: no barrier in front of the writes.)
        OfxCand fullbest = ofx_cand_none();
        {
            const float wmax = ofx_wave_max(mloc);
            if (mloc == wmax) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (__builtin_amdgcn_ballot_w64(gm[g] == wmax) == 0) continue;    // uniform
                    const int h = (GS * g) / R1;
                    if (h == NRF && !w2i) continue;                                   // zeros
                    const int base = 2 * (h == NRF ? vt3t : tct + GT * h) + pre;
#pragma unroll
                    for (int j = GS * g; j < GS * g + GS; ++j) {
                        const int n1 = j - R1 * h;
                        const cpx v = d[j];
                        int i0 = base + ROWS * n1;
                        i0 = i0 >= GN ? i0 - GN : i0;
                        int i1 = i0 + 1;
                        i1 = i1 >= GN ? i1 - GN : i1;
                        if (v.x * v.x == wmax && i0 < fullbest.idx) {
                            fullbest.idx = i0; fullbest.amp = v.x; fullbest.key = wmax;
                        }
                        if (v.y * v.y == wmax && i1 < fullbest.idx) {
                            fullbest.idx = i1; fullbest.amp = v.y; fullbest.key = wmax;
                        }
                    }
                }
            }
            fullbest = ofx_cand_wave_reduce(fullbest);
            const float wchi = ofx_wave_sum(chi0p);
            if (lane_t == 0) {
                L.cand[wave_t] = fullbest;
                L.red[1][wave_t] = wchi;
            }
            if (tt == 0) L.bcast[0] = d[0].x;          // A(lag 0)
            __syncthreads();
        }
        float chi0 = L.red[1][0];
        fullbest = L.cand[0];
#pragma unroll
        for (int q = 1; q < NWAVE; ++q) {
            const OfxCand o = L.cand[q];
            if (ofx_cand_better(o.key, o.idx, fullbest)) fullbest = o;
            chi0 += L.red[1][q];
        }
        float a_lag0 = L.bcast[0];
        asm volatile("" : "+v"(chi0), "+v"(a_lag0));
        SUBSTAMP(13);
        SUBSTAMP(14);

        // windowed / outside-window fits scan the lag dump: even lags, then odd lags
        if constexpr (FEAT & 1) {
            if (lane_t < OFX_MAX_SEARCHES) L.wc[lane_t][wave_t] = ofx_cand_none();
            // Narrow windows (the usual case: +-100 us is 250 lags, one or two of the R1 register rows
            // of 1250 lags): only the rows the windowed searches touch are dumped, complex, in ONE
            // pass (row n1 -> slot popcount(rowmask below n1); up to NSLOT rows fit the buffer).
            const unsigned rmask = TBX.rowmask;
            constexpr int NSLOT = (2 * XB_ELEMS) / ROWS;
            if (__builtin_popcount(rmask) <= NSLOT) {
                __syncthreads();
                int slotb = 0;                                   // uniform: slot * 625
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) {
                    if (!((rmask >> n1) & 1u)) continue;         // uniform
#pragma unroll
                    for (int h = 0; h < NR1; ++h) {
                        if (h == NRF && !w2i) continue;
                        const int vth = (h == NRF) ? vt3t : tct + GT * h;
                        xc[slotb + vth] = d[R1 * h + n1];
                    }
                    slotb += NV1;
                }
                __syncthreads();
#pragma unroll 1
                for (int q = 0; q < SDX.n_search; ++q) {
                    const OfxSearchDev& sq = SDX.search[q];
                    const bool full = !sq.outside && sq.lo == 0 && sq.hi == GN;
                    if (sq.kind != OFX_SEARCH_DELAY || full) continue;
                    OfxCand c = ofx_cand_none();
                    auto scan = [&](int i0, int i1) {
                        for (int i = i0 + tt; i < i1; i += BLK) {
                            int n = i - pre;
                            n = n < 0 ? n + GN : n;
                            const int n1 = n / ROWS;
                            const int sl = __builtin_popcount(rmask & ((1u << n1) - 1u));
                            ofx_cand_take(c, SH.xb[sl * ROWS + (n - ROWS * n1)], i);
                        }
                    };
                    if (sq.outside) {
                        scan(0, sq.lo);
                        scan(sq.hi, GN);
                    } else {
                        scan(sq.lo, sq.hi);
                    }
                    c = ofx_cand_wave_reduce(c);
                    if (lane_t == 0) L.wc[q][wave_t] = c;
                }
            } else
            for (int e = 0; e < 2; ++e) {
                __syncthreads();
#pragma unroll
                for (int h = 0; h < NR1; ++h) {
                    if (h == NRF && !w2i) continue;
                    const int vth = (h == NRF) ? vt3t : tct + GT * h;
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1)
                        SH.xb[NV1 * n1 + vth] = e ? d[R1 * h + n1].y : d[R1 * h + n1].x;
                }
                __syncthreads();
#pragma unroll 1
                for (int q = 0; q < SDX.n_search; ++q) {
                    const OfxSearchDev& sq = SDX.search[q];
                    const bool full = !sq.outside && sq.lo == 0 && sq.hi == GN;
                    if (sq.kind != OFX_SEARCH_DELAY || full) continue;
                    OfxCand c = ofx_cand_none();
                    auto scan = [&](int i0, int i1) {
                        for (int i = i0 + tt; i < i1; i += BLK) {
                            int n = i - pre;
                            n = n < 0 ? n + GN : n;
                            if ((n & 1) == e) ofx_cand_take(c, SH.xb[n >> 1], i);
                        }
                    };
                    if (sq.outside) {
                        scan(0, sq.lo);
                        scan(sq.hi, GN);
                    } else {
                        scan(sq.lo, sq.hi);
                    }
                    c = ofx_cand_wave_reduce(c);
                    if (lane_t == 0 && ofx_cand_better(c.key, c.idx, L.wc[q][wave_t]))
                        L.wc[q][wave_t] = c;
                }
            }
            __syncthreads();
        }

        // psd_amp bands from the stashed 2 X_k; one wave per band
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rxw = make_rsrc(nullptr, 0);   // (no global stash at these lengths)
#include "ofx_fused_bands.inc"

#include "ofx_fused_resolve.inc"
        // interpolate=True: amplitudes at the rolled bins idx -+ 1
        if constexpr (FEAT & 1) {
#pragma unroll 1
            for (int q = 0; q < SDX.n_search; ++q) {
                const OfxSearchDev& sq = SDX.search[q];
                if (!sq.interp) continue;
                const OfxCand best = resolve(sq, q);
                // lag n = 1250 n1 + 2 vt + e sits in thread vt % 250, register 20 (vt / 250) + n1
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    int n = best.idx + (side ? 1 : -1) - pre;
                    n = n < 0 ? n + GN : (n >= GN ? n - GN : n);
                    const int n1 = n / ROWS;
                    const int vt_n = (n - ROWS * n1) >> 1;
                    const int hh = vt_n / GT;
                    if (best.idx != 0x7fffffff && tt == vt_n - GT * hh + (hh == NRF ? I3T : 0)) {
                        const int jn = R1 * hh + n1;
                        cpx v = d[0];
#pragma unroll
                        for (int j = 1; j < NV; ++j) v = (j == jn) ? d[j] : v;
                        L.nb[q][side] = (n & 1) ? v.y : v.x;
                    }
                }
                __syncthreads();
                const OfxRefined ref = ofx_interpolate(L.nb[q][0], best.amp, L.nb[q][1], best.idx,
                                                       GN, SDX.norm, chi0);
                if (tt == 0) L.ref[q] = ref;
            }
            __syncthreads();
        }
        SUBSTAMP(15);
        // d and the time-domain temporaries are dead: the request of the next trace (or of the
        // spectrum again, for the next slot)
        // (one request site, ahead of the low-frequency chi2: its HBM latency hides under the terms
        // below; the tables of those terms were requested before it, so that waiting for them --
        // vmcnt counts in order -- does not wait for the trace)
        if (MULTI && slot_i + 1 < slot_n) {
            load_spec();
        } else {
            const long long bn = b + stride;
            have = bn < n_traces;
            if (have) load_trace(bn);
        }
        {
#pragma unroll 1
            for (int q = 0; q < SDX.n_search; ++q) {
                const OfxSearchDev& sq = SDX.search[q];
                const OfxCand best = resolve(sq, q);
                const int dl = best.idx - pre;
                OfxRefined ref;
                ref.amp = best.amp;
                ref.frac = 0.0f;
                ref.chi2 = 0.0f;
                if constexpr (FEAT & 1)
                    if (sq.interp) ref = L.ref[q];
                float low = 0.0f;
                cpx ph = lowchi2_phase(act_t ? tt : 0, dl, ref.frac);
                const cpx step = lowchi2_phase(GT, dl, ref.frac);        // uniform
#pragma unroll
                for (int i = 0; i < NLK; ++i) {
                    const int k = tt + GT * i;
                    if (act_t && k < sq.nlow)
                        low += lowchi2_term(k, ph, ref.amp, L.xlow[k], lk_s[i], lk_g[i]);
                    ph = cmul(ph, step);
                }
                low = ofx_wave_sum(low);
                if (lane_t == 0) L.lowp[q][wave_t] = low;
                if (tt == 0) L.fin[q] = best;
            }
            STAMP(11);
            __syncthreads();
#include "ofx_fused_row_write.inc"
        }
        }
#undef SDX
#undef TBX
#if OFX_TPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(12);
    }
#ifdef OFX_STAMPS
    asm volatile("s_dcache_wb");
#endif
}


// ------------------------------------------------------------------ the transform on its own
// Batched complex transform of rows of GM points, natural order in and out, unnormalised (the
// interface of ofx_ldsfft_exec, ofx_lds.hip): the register-resident stages of k_fused25 without
// the middle step and the tail.  It serves the N x M engine and the ROCFFT engine at these
// trace lengths (their transforms run on the GM packed points of a trace), in place of the
// LDS-resident transform k_lds_fft.
//   forward: rows z[m] -> Z[k] = sum_m z[m] exp(-2 pi i m k / GM);  inverse: the + sign.
template <bool FWD>
__global__ __launch_bounds__(BLK, WG_PER_CU * NWAVE / 4) void k_fft25(
    const float2* __restrict__ t1, const float2* __restrict__ t2tab, const float2* __restrict__ in,
    float2* __restrict__ out, long long rows) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Shared25& SH = *reinterpret_cast<Shared25*>(smem_raw);
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool w2 = PART && wave < W3;
    const __amdgpu_buffer_rsrc_t t1q = make_rsrc(t1, 2 * T1PAD * 16);
    for (int i = tid; i < R2 * R3; i += BLK) SH.t2[i] = mk(t2tab[i].x, t2tab[i].y);
    cpx d[NV], nd[NV];
    cpx* const xc = reinterpret_cast<cpx*>(SH.xb);
    auto pass2 = [](int h, int k2) { return k2 < 12 ? 0 : (k2 > 12 ? 1 : h); };
    for (long long b = (long long)blockIdx.x; b < rows; b += (long long)gridDim.x) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int tc = min(tl, GT - 1);
        const int vt3 = min(tl + NRF * GT, NV1 - 1);
        const int k1l = (tc * 1311) >> 15;
        const int n3u = tc - 25 * k1l;
        const int rb1 = k1l * LD1 + n3u;
        const int rbB = (tc == 0) ? 0 : (GT - tc);
        const int rowB = (tc == 0) ? GP / 2 : GP - tc;      // partner block of thread v
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)b * GM, GM * 8);
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(out + (size_t)b * GM, GM * 8);
        auto st2 = [&](cpx v, int idx) {
            u32x2 u;
            u.x = __float_as_uint(v.x);
            u.y = __float_as_uint(v.y);
            __builtin_amdgcn_raw_buffer_store_b64(u, rout, idx * 8, 0, 0);
        };
        if constexpr (FWD) {
#pragma unroll
            for (int h = 0; h < NRF; ++h)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1)
                    d[R1 * h + n1] = buf_ld2(rin, (tc + GT * h) * 8, n1 * NV1 * 8);
            if (w2) {
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) d[NRF * R1 + n1] = buf_ld2(rin, vt3 * 8, n1 * NV1 * 8);
            }
            {
                T1Anch gf[NRF];
#pragma unroll
                for (int h = 0; h < NRF; ++h) gf[h] = t1_load(t1q, tc + GT * h);
                r1_dfts<-1>(d);
                r1_twiddles<false>(d, gf);
                if (w2) {
                    const T1Anch g2 = t1_load(t1q, vt3);
                    dft_r1<-1, (NR1 - 1) * R1>(d);
                    t1_apply<false, (NR1 - 1) * R1>(d, g2);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {                       // E1
                __syncthreads();
#pragma unroll
                for (int h = 0; h < NRF; ++h)
#pragma unroll
                    for (int j = 0; j < HR1; ++j) xc[j * LD1 + tc + GT * h] = d[R1 * h + HR1 * p + j];
                if (w2) {
#pragma unroll
                    for (int j = 0; j < HR1; ++j) xc[j * LD1 + vt3] = d[(NR1 - 1) * R1 + HR1 * p + j];
                }
                __syncthreads();
#pragma unroll
                for (int n2 = 0; n2 < R2; ++n2) nd[R2 * p + n2] = xc[rb1 + R3 * n2];
            }
#pragma unroll
            for (int j = 0; j < 2 * R2; ++j) d[j] = nd[j];
            dft25<-1, NV, 0>(d);                                // F2
            dft25<-1, NV, R2>(d);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 1; k2 < R2; ++k2)
                    d[R2 * h + k2] = cmul(d[R2 * h + k2], SH.t2[k2 * R3 + n3u]);
#pragma unroll
            for (int p = 0; p < 2; ++p) {                       // E2
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k2 = 0; k2 < R2; ++k2)
                        if (pass2(h, k2) == p)
                            xc[(k1l + HR1 * h + R1 * k2 - GT * p) * R3 + n3u] = d[R2 * h + k2];
                __syncthreads();
                const int rr = (p == 0) ? tc : rbB;
#pragma unroll
                for (int j = 0; j < R3; ++j) nd[R3 * p + j] = xc[rr * R3 + j];
            }
#pragma unroll
            for (int j = 0; j < 2 * R3; ++j) d[j] = nd[j];
            dft25<-1, NV, 0>(d);                                // F3: blocks v and 500 - v
            dft25<-1, NV, R3>(d);
#pragma unroll
            for (int j = 0; j < R3; ++j) {                      // Z[k_low + 500 k3], natural order
                st2(d[j], tc + GP * j);
                st2(d[R3 + j], rowB + GP * j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < R3; ++j) {
                d[j] = buf_ld2(rin, tc * 8, j * GP * 8);
                d[R3 + j] = buf_ld2(rin, rowB * 8, j * GP * 8);
            }
            dft25<+1, NV, 0>(d);                                // I3
            dft25<+1, NV, R3>(d);
#pragma unroll
            for (int p = 0; p < 2; ++p) {                       // E3
                __syncthreads();
                {
                    const int rr = (p == 0) ? tc : rbB;
#pragma unroll
                    for (int j = 0; j < R3; ++j) xc[rr * R3 + j] = d[R3 * p + j];
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k2 = 0; k2 < R2; ++k2)
                        if (pass2(h, k2) == p)
                            nd[R2 * h + k2] = xc[(k1l + HR1 * h + R1 * k2 - GT * p) * R3 + n3u];
            }
#pragma unroll
            for (int j = 0; j < 2 * R2; ++j) d[j] = nd[j];
#pragma unroll
            for (int h = 0; h < 2; ++h)                         // I2
#pragma unroll
                for (int k2 = 1; k2 < R2; ++k2)
                    d[R2 * h + k2] = cmulc(d[R2 * h + k2], SH.t2[k2 * R3 + n3u]);
            dft25<+1, NV, 0>(d);
            dft25<+1, NV, R2>(d);
            T1Anch gi[NRF];
#pragma unroll
            for (int h = 0; h < NRF; ++h) gi[h] = t1_load(t1q, tc + GT * h);
#pragma unroll
            for (int p = 0; p < 2; ++p) {                       // E4
                __syncthreads();
#pragma unroll
                for (int n2 = 0; n2 < R2; ++n2) xc[rb1 + R3 * n2] = d[R2 * p + n2];
                __syncthreads();
#pragma unroll
                for (int h = 0; h < NRF; ++h)
#pragma unroll
                    for (int j = 0; j < HR1; ++j) nd[R1 * h + HR1 * p + j] = xc[j * LD1 + tc + GT * h];
                if (p == 0 && w2) {
#pragma unroll
                    for (int j = 0; j < HR1; ++j) nd[(NR1 - 1) * R1 + j] = xc[j * LD1 + vt3];
                }
            }
#pragma unroll
            for (int j = 0; j < NRF * R1; ++j) d[j] = nd[j];
            r1_twiddles<true>(d, gi);                           // I1
            r1_dfts<+1>(d);
#pragma unroll
            for (int h = 0; h < NRF; ++h)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) st2(d[R1 * h + n1], NV1 * n1 + tc + GT * h);
            if constexpr (PART) {
                if (w2) {
                    const T1Anch g2 = t1_load(t1q, vt3);
#pragma unroll
                    for (int j = 0; j < HR1; ++j) {
                        d[NRF * R1 + j] = nd[NRF * R1 + j];
                        d[NRF * R1 + HR1 + j] = xc[j * LD1 + vt3];
                    }
                    t1_apply<true, NRF * R1>(d, g2);
                    dft_r1<+1, NRF * R1>(d);
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) st2(d[NRF * R1 + n1], NV1 * n1 + vt3);
                }
            }
        }
    }
}

}  // namespace

// =============================================================== host side
bool OFX25_FN(supported)(int n_samples) { return n_samples == GN; }

static int fused25_tables(ofx_plan* p) {
    if (p->d_tw1) return OFX_OK;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<float2> t1(4 * T1PAD, make_float2(1.0f, 0.0f)), t2(R2 * R3 + VPAD);
    const int anchor_mult[4] = {1, 2, 5, 10};                 // B1, B2, A1, A2
    for (int i = 0; i < 4; ++i)
        for (int n = 0; n < T1PAD; ++n) {
            const long long e = ((long long)anchor_mult[i] * n) % GM;
            const double a = -PI2 * (double)e / GM;
            t1[((i >> 1) * T1PAD + n) * 2 + (i & 1)] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k2 = 0; k2 < R2; ++k2)
        for (int n3 = 0; n3 < R3; ++n3) {
            const int e = (k2 * n3) % (R2 * R3);
            const double a = -PI2 * (double)e / (double)(R2 * R3);
            t2[k2 * R3 + n3] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int v = 0; v < VPAD; ++v) {      // tbase[v] = i exp(-2 pi i v / N), stored after t2
        const double a = -PI2 * (double)v / GN;
        t2[R2 * R3 + v] = make_float2((float)-std::sin(a), (float)std::cos(a));
    }
    return fused_upload_tables(p, t1, t2);
}

// Middle-step tables of one slot from the fp64 one-sided filter.
//   d_pq (float4 units): [0 .. 25*256) midW (W_k / 2, conj(W_p) / 2) [slot J][v];
//   [25*256 .. +25*128) midG (g_k', g_p') as float2 [slot J][v]; last entry (W_{M/2}, g_{M/2}, 0).
// Slot J of thread v pairs bin k = v + 500 J with p = M - k (v = 0: k = 500 J for J <= 12 and
// 250 + 500 (J - 13) above; k = 0 pairs DC with Nyquist).
int OFX25_FN(prepare_slot)(ofx_plan* p, int slot, const double* wf) {
    int rc = fused25_tables(p);
    if (rc) return rc;
    return fused_build_slot_tables(p, slot, wf, GM, GT, VPAD, R3, [](int v, int j) {
        return v != 0 ? v + GP * j : (j <= 12 ? GP * j : GP / 2 + GP * (j - 13));
    });
}

template <int FEAT, bool MULTI>
static int launch25(ofx_plan* p, const OfxPlanDev& pd, const OfxSlotDev& sd, const Tabs25& tabs,
                    const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                    hipStream_t st, const SlotArg25* d_slots, int nslots) {
    OFX_LDS_ATTR_ONCE((k_fused25<FEAT, MULTI>), LDS_BYTES);
    long long grid = (long long)p->cu_count * WG_PER_CU;
    if (MULTI) {
        const size_t need = (size_t)p->cu_count * WG_PER_CU * 2 * R3 * BLK * sizeof(float2);
        if (int rcs = fused_ensure_spec(p, need)) return rcs;
    }
    if (grid > n) grid = n;
#ifdef OFX_STAMPS
    const size_t stamp_bytes = (size_t)grid * OFX_STAMP_TRACES * NWAVE * 16 * sizeof(unsigned long long);
    if (p->d_fused_spec) (void)hipFree(p->d_fused_spec);
    p->d_fused_spec = nullptr;
    p->fused_spec_bytes = 0;
    OFX_HIP(hipMalloc(&p->d_fused_spec, stamp_bytes));
    OFX_HIP(hipMemset(p->d_fused_spec, 0, stamp_bytes));
#endif
    size_t tix = 0;
    int rc = ofx_time_begin(p, st, &tix);
    if (rc) return rc;
    hipLaunchKernelGGL((k_fused25<FEAT, MULTI>), dim3((unsigned)grid), dim3(BLK), LDS_BYTES, st, pd,
                       sd, tabs, d_traces, d_valid, n, d_out, d_slots, nslots,
                       reinterpret_cast<float2*>(p->d_fused_spec));
    rc = ofx_time_end(p, st, tix);
    if (rc) return rc;
    OFX_HIP(hipGetLastError());
#ifdef OFX_STAMPS
    if (int rcd = fused_dump_stamps(st, p->d_fused_spec, stamp_bytes)) return rcd;
#endif
    return OFX_OK;
}

template <bool MULTI>
static int launch25_feat(int feat, ofx_plan* p, const OfxPlanDev& pd, const OfxSlotDev& sd,
                         const Tabs25& tabs, const float* d_traces, const uint8_t* d_valid,
                         long long n, float* d_out, hipStream_t st, const SlotArg25* d_slots,
                         int nslots) {
#ifdef OFX_QUICK      // development builds: only the headline variant is compiled
    if (feat == 0 && !MULTI)
        return launch25<0, false>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots, nslots);
    ofx_set_error("OFX_QUICK build: only the FEAT = 0 single-slot kernel exists");
    return OFX_ERR_UNSUPPORTED;
#else
#define OFX_CASE(F)                                                                            \
    case F:                                                                                    \
        return launch25<F, MULTI>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,   \
                                  nslots);
    switch (feat & 7) {
        OFX_CASE(0) OFX_CASE(1) OFX_CASE(2) OFX_CASE(3) OFX_CASE(4) OFX_CASE(5) OFX_CASE(6)
        default: return launch25<7, MULTI>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st,
                                           d_slots, nslots);
    }
#undef OFX_CASE
#endif
}

int OFX25_FN(process)(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n,
                        float* d_out, hipStream_t st) {
    int rc = fused25_tables(p);
    if (rc) return rc;
    Tabs25 common;
    memset(&common, 0, sizeof(common));
    common.t1 = p->d_tw1;
    common.t2 = p->d_tw2;
    common.tbase = p->d_tw2 + R2 * R3;
    {
        // thread 0, slots J >= 13: bin 250 + 500 (J - 13) = 500 J + (250 - 6500)
        const double a = -6.283185307179586476925286766559 * (double)(GP / 2 - 13 * GP) / GN;
        common.tb0hi = make_float2((float)-std::sin(a), (float)std::cos(a));
    }
    common.midW = reinterpret_cast<const float4*>(p->d_tw1);
    common.midG = p->d_tw1;

    struct G {
        enum { N = GN, ROWS = 2 * NV1, NROWS = R1, LDS_BINS = NLOW_MAX, MAX_BINS = NLOW_MAX,
               MIDG_OFF = R3 * VPAD };
        using Tabs = Tabs25;
        using SlotArg = SlotArg25;
    };
    return fused_process_plan<G>(p, common, st, [&](bool multi, int feat, const OfxPlanDev& pd,
                                                    const OfxSlotDev& sd, const Tabs25& tabs,
                                                    const SlotArg25* d_slots, int nslots, int) {
        return multi ? launch25_feat<true>(feat, p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,
                                           nslots)
                     : launch25_feat<false>(feat, p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,
                                            nslots);
    });
}

// ---- the transform on its own (ofx_ldsfft_* dispatch to these for GM-point rows)
int OFX25_FN(fft_create)(int n_complex, int device, void** out) {
    if (n_complex != GM) return OFX_ERR_UNSUPPORTED;
    return fused_fft_create(device, out, fused25_tables);
}
void OFX25_FN(fft_destroy)(void* h) { fused_fft_destroy(h); }
int OFX25_FN(fft_exec)(void* h, bool forward, const float2* in, float2* out, long long rows,
                       hipStream_t st) {
    return fused_fft_exec<&k_fft25<true>, &k_fft25<false>>(h, forward, in, out, rows, st, WG_PER_CU, BLK,
                                                           sizeof(Shared25));
}
