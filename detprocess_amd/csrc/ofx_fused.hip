// ofx_fused.hip -- FUSED engine: a persistent workgroup carries a whole trace through
//   load -> real FFT -> optimal filter + chi2_0 -> inverse FFT -> arg-max / chi2 /
//   low-frequency chi2 -> one output row,   touching HBM once per sample
// (SURVEY.md section 7 step 5).
//
// Geometry for N = 32768 real samples (M = N/2 = 16384 packed complex points,
// z[m] = x[2m] + i x[2m+1]); the whole trace lives in registers, M = 32 x 32 x 16:
//   m = 512 n1 + 16 n2 + n3      k = k1 + 32 k2 + 1024 k3
//   F1: virtual thread t = n' = 16 n2 + n3, radix-32 over n1 -> k1, x w_M^{n' k1}
//   E1: LDS exchange D1[k1][n']            (row stride 528: conflict-free)
//   F2: virtual thread u = 16 k1 + n3, radix-32 over n2 -> k2, x w_512^{n3 k2}
//   E2: LDS exchange D2[k1 + 32 k2][n3]    (row stride 17: conflict-free)
//   F3: virtual thread v owns the two 16-point blocks k_low = v and its Hermitian
//       partner 1024 - v (v = 0: the two self-paired blocks 0 and 512), so the
//       real-FFT unpack, the filter multiply, chi2_0 and the re-pack for the
//       inverse need NO exchange:  (Z_k, Z_{M-k}) live in the same thread.
//   I3 / E3 / I2 / E4 / I1 mirror F3 / E2 / F2 / E1 / F1 with conjugate twiddles.
// After I1 virtual thread t holds A(n) for the 64 lags n = 1024 n1 + 2 t + {0,1}.
//
// 512 virtual threads x 32 complex values.  A hardware thread carries VT of them
// (VT = 2: 256-thread workgroups, 64 complex = 128 VGPRs of data, two workgroups
// resident per CU so that one computes while the other sits in LDS / barrier / HBM
// waits, and every thread has two independent instruction streams).
//
// Arithmetic is packed fp32 (ofx_fft_regs.h: a complex value = one aligned VGPR pair, one
// v_pk_* instruction per complex operation).  Stage-1 twiddles come from ten per-thread
// anchors, the middle-step twiddles from a per-thread base times compile-time constants;
// only the filter (24 bytes per bin pair) is streamed from L2.  Kernel variants:
//   FEAT  bit 0 windowed / interpolating searches (lags dumped to LDS), bit 1 time-domain
//         windows, bit 2 channel algebra on load
//   MULTI several filter slots share the forward transform (spectrum parked per workgroup)
// Build-time switches (diagnostics, see tools/README.md): OFX_VT / OFX_WGPC (128-VGPR,
// four-waves-per-SIMD layout), OFX_MID_DEPTH, ABL_NOEXCH / ABL_NOTAB / ABL_NOTAIL / ABL_LOADONLY.
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

#ifndef OFX_VT
#define OFX_VT 2
#endif
constexpr int VT = OFX_VT;         // virtual threads per hardware thread
constexpr int FN = 32768;          // samples
#define GEO_N FN                   // (the names the shared fragments ofx_fused_*.inc use)
#define GEO_WIDE WIDE
#define GEO_LDS_BINS NLOW_MAX
constexpr int FM = 16384;          // packed complex points
constexpr int FV = 512;            // virtual threads
constexpr int FT = FV / VT;        // hardware threads per workgroup
// PP ("ping-pong", build with -DOFX_PP=1; measured and NOT the product configuration): ONE
// 512-thread workgroup per CU carries TWO traces, one per half (waves 0-3 / 4-7: the two waves
// of every SIMD belong to different halves).  The halves run the same program two barriers
// apart, so that whenever one half is in an LDS exchange the other is in a register-arithmetic
// phase: the exchange buffer is shared (full size: every exchange is a single pass) and never
// has two users.  Motivation: with two independent 256-thread workgroups per CU both sit in
// exchange phases at the same time more than half of the time
// (profiles/r02_phase_timeline_before.json).  Result (profiles/r02_phase_timeline_pingpong.json):
// correct (every GPU test passes) but 6 % slower, 14.6 against 15.6 M traces/s -- in lock-step
// every stall of one half (filter-table latency in the middle step, the reductions of the
// tail, the wait for the next trace) holds the partner at the next barrier, whereas independent
// workgroups fill each other's stalls.  Kept as a build option and as the record of that
// experiment.
#ifndef OFX_PP
#define OFX_PP 0
#endif
constexpr bool PP = (OFX_PP != 0) && (OFX_VT == 2);
#ifndef OFX_WGPC
#define OFX_WGPC OFX_VT
#endif
constexpr int WG_PER_CU = PP ? 1 : OFX_WGPC;    // workgroups resident per CU
constexpr int NHALF = PP ? 2 : 1;               // traces in flight per workgroup
constexpr int BLOCK_THREADS = FT * NHALF;
constexpr int NV = 32 * VT;        // complex values per hardware thread
constexpr int LD1 = 528;           // D1 row stride (elements); 528 = 16 mod 32
constexpr int LD2 = 17;            // D2 row stride (elements)
constexpr int XBUF_ELEMS = 1024 * LD2;            // 17408 >= 32*513, >= 16384
constexpr int NLOW_MAX = 512;         // low bins 2 X_k kept in LDS
constexpr int NS_MAX = 4096;           // ... and up to here in the workgroup's global stash (WIDE)
constexpr int NWAVE = FT / OFX_WAVE;

// Wave priorities by phase (s_setprio).  The two workgroups of a CU put one wave each on
// every SIMD; at equal priority the older wave wins every issue tie, so the younger workgroup's
// latency chains (the write -> barrier -> read -> barrier of an exchange, the reductions of the
// tail, the table-fed middle step) queue behind the older one's butterflies.  Ranking the
// phases instead -- exchanges 2, middle step and tail 1, the DFT blocks 0 -- lets a latency
// chain issue at once and costs the arithmetic almost nothing: 15.7 -> 17.4 M traces/s on
// the same box (+11 %; other rankings: 1/1/1 +8 %, 2/1/0 +4 %, 3/0/0 +2 %).
#ifndef OFX_XPRIO
#define OFX_XPRIO 2
#endif
#ifndef OFX_TPRIO
#define OFX_TPRIO 1
#endif
#ifndef OFX_MPRIO
#define OFX_MPRIO 1
#endif

// With two workgroups per CU every exchange runs in two passes through a half-size
// buffer (68 KiB per workgroup instead of 136): pass 0 moves the rows of the lower
// half of the layout (D1: k1 < 16, D2: k_low < 512), pass 1 the upper half.  Values
// stay complex (ds_write_b64 / ds_read_b64: 2/3 of the LDS cycles of a re/im split).
constexpr bool SPLIT_EXCHANGE = !PP && (WG_PER_CU > 1);
constexpr int HB1 = 16 * LD1;      // D1 elements per half
constexpr int HB2 = 512 * LD2;     // D2 elements per half

struct FusedShared {               // one per workgroup
    float xb[SPLIT_EXCHANGE ? XBUF_ELEMS : 2 * XBUF_ELEMS];   // exchange buffer / lag dump
    cpx t2[512];                   //  4,096 B   w_512^{n3 k2}, index k2*16+n3
};
struct FusedLds {                  // one per trace in flight (per half)
    cpx xlow[NLOW_MAX + 8];        //  4,160 B   2*X_k for k < 512 (lowchi2)
    float red[4][NWAVE];           // per-wave partials
    // time-domain window partials, double-buffered by trace parity: a plan with windows and no
    // search has no barrier between thread w's reads of one trace and the other waves' writes of
    // the next (a barrier on that path costs every kernel with windows ~420 B of scratch per lane:
    // it splits the region the prefetched trace lives in)
    float tdred[2][OFX_MAX_TDWIN][4][NWAVE];
    OfxCand cand[NWAVE];
    OfxCand wc[OFX_MAX_SEARCHES][NWAVE];    // per-wave winners of the windowed searches
    OfxCand fin[OFX_MAX_SEARCHES];          // resolved fit per search
    float lowp[OFX_MAX_SEARCHES][NWAVE];    // low-frequency chi2 per wave
    float bcast[8];
    cpx perm[32];                           // virtual thread 0's permutation bounce buffer
    float nb[OFX_MAX_SEARCHES][2];          // amplitudes next to the winner (interpolation)
    OfxRefined ref[OFX_MAX_SEARCHES];       // refined fits
};
constexpr size_t FUSED_LDS_BYTES = sizeof(FusedShared) + NHALF * sizeof(FusedLds);
static_assert(FUSED_LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS budget");

#include "ofx_fused_stamps.h"

struct FusedTabs {
    const float2* t1;     // [5][512] float4 rows of stage-1 twiddle anchors (see T1Anch)
    const float2* t2;     // [32][16]   w_512^{n3 k2}
    const float4* midW;   // [16][512]  (W_k / 2, conj(W_p) / 2)          slot j, virtual thread v
    const float2* midG;   // [16][512]  (g_k', g_p')
    const float2* tbase;  // [512]      T_v = i exp(-2 pi i v / N); T of slot j is T_v w_32^j
    float2 tb0hi;         // base of virtual thread 0 for its slots j >= 8 (block 512); read from the kernel ARGUMENT only (ofx_fused25.hip, Tabs25::tb0hi)
    float2 wq;            // W_{M/2}  (the self-paired bin k = M/2)
    float gq;             // g_{M/2}
    unsigned rowmask;     // register rows n1 (1024 lags each) the slot's windowed searches touch
};

// One filter slot of a multi-slot launch (device array, read through scalar loads).
struct FusedSlotArg {
    OfxSlotDev sd;
    FusedTabs tabs;
};

__device__ __forceinline__ int partner_block(int v) { return v == 0 ? 512 : 1024 - v; }

#include "ofx_fused_parts.h"
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


// Middle step over the 32 values of a virtual thread at d[O .. O+32).
// A = d[O+0..15] (block k_low = v), B = d[O+16..31] (partner block).  Generic: slot j
// pairs (A[j], B[15-j]).  Virtual thread 0 (blocks 0 and 512, both self-paired) is
// brought to the same slot shape by a permutation of its 32 values (perm_in / perm_out):
// that one thread bounces them through 256 bytes of LDS, which costs no registers.
//   genA = [A0[0..7], B0[0..7]] ; genB = [B0[8..15], A0[9..15], A0[0]]
__device__ constexpr int perm_in_src(int j) {
    if (j < 8) return j;
    if (j < 24) return j + 8;                      // genA[8..15] = B0[0..7]; genB[0..7] = B0[8..15]
    return (j - 16 + 1) % 16;                      // genB[8..15] = A0[9..15], A0[0]
}
// A0[9..15] = genB[8..14], A0[0] = genA[0] (genB[15] discarded); B0[0..7] = genA[8..15];
// B0[8..15] = genB[0..7]; A0[8] is the self-paired bin k = M/2, handled by the caller.
__device__ constexpr int perm_out_src(int j) {
    if (j < 8) return j;
    if (j < 16) return 16 + j - 1;                 // j = 8 is overwritten by the caller
    if (j < 24) return j - 8;
    return j - 8;
}
template <int O>
__device__ __forceinline__ void perm_in(cpx (&d)[NV], bool z, cpx* buf) {
    if (z) {
#pragma unroll
        for (int j = 0; j < 32; ++j) buf[j] = d[O + j];
#pragma unroll
        for (int j = 8; j < 32; ++j) d[O + j] = buf[perm_in_src(j)];
    }
}
template <int O>
__device__ __forceinline__ cpx perm_out(cpx (&d)[NV], bool z, cpx a8, const FusedTabs& tabs,
                                        cpx chi, cpx* buf) {
    if (z) {
        // self-paired bin k = M/2 (A0[8]):  X = conj(Z), Z' = 2 conj(W) Z
        const cpx zq = cmulc(a8, mk(tabs.wq.x, tabs.wq.y));
        // (2 g from the scalar argument, behind an opaque copy: as a loop-invariant vector pair it was
        // spilled before the loop and reloaded here)
        int gqi = __float_as_int(tabs.gq);
        asm volatile("" : "+s"(gqi));
        const float g2 = 2.0f * __int_as_float(gqi);
        chi = pfma(a8 * a8, mk(g2, g2), chi);
#pragma unroll
        for (int j = 0; j < 32; ++j) buf[j] = d[O + j];
#pragma unroll
        for (int j = 9; j < 32; ++j) d[O + j] = buf[perm_out_src(j)];
        d[O + 8] = zq + zq;
    }
    return chi;
}

// 16 pair slots; table rows are software-pipelined MID_DEPTH slots ahead (L2 latency is
// ~1k cycles against ~100 cycles of arithmetic per slot).  The twiddle of slot j is the
// thread's base T_v times the constant w_32^j: only the filter itself comes from memory
// (24 bytes per slot).  tlo / thi: base for slots j < 8 / j >= 8 (they differ only in
// virtual thread 0, whose upper slots belong to block 512).
#ifndef OFX_MID_DEPTH
#define OFX_MID_DEPTH 4
#endif
constexpr int MID_DEPTH = OFX_MID_DEPTH;
struct MidRsrc {
    __amdgpu_buffer_rsrc_t w, g;
    __amdgpu_buffer_rsrc_t xw;     // WIDE: this workgroup's stash of 2 X_k, k = 512 .. nstash-1
};
// WIDE kernels also keep the bins k >= NLOW_MAX a low-frequency chi2 cut-off or a psd_amp band
// reaches (the reference's example YAML uses lowchi2_fcutoff = 50 kHz: 1311 bins at 32768
// samples, examples/processing/process_example.yaml:113) in a per-workgroup stash in global
// memory (L2-resident, entry k - 512).  Who holds bin k:  slot J of virtual thread v has
// xk2 = 2 X_k for k = v + 1024 J and xp2 = 2 conj(X_p) for p = 1024 (16 - J) - v (v != 0);
// virtual thread 0's slots J >= 8 hold block 512: k = 512 + 1024 (J - 8).  The descriptor is
// sized to the bins the plan needs (and empty for the slots after the first of a MULTI
// launch): stores beyond it are dropped by the bounds check, so there are no branches here.
// The B-side entries stay conjugated; the reader undoes that ((k & 1023) > 512).
constexpr int OOB_OFF = 0x40000000;
__device__ __forceinline__ void buf_st2(__amdgpu_buffer_rsrc_t r, cpx x, int voff) {
    u32x2 v;
    v.x = __float_as_uint(x.x);
    v.y = __float_as_uint(x.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}
template <int O, int J, int NB, bool WIDE>
__device__ __forceinline__ void mid_unrolled(cpx (&d)[NV], const MidRsrc& r, int v, FusedLds& L,
                                             cpx tlo, cpx thi, float4 (&tw)[NB], cpx (&tg)[NB],
                                             cpx& chi) {
    if constexpr (J < 16) {
        if constexpr (J + MID_DEPTH < 16) {
            tw[(J + MID_DEPTH) % NB] = buf_ld4(r.w, v * 16, (J + MID_DEPTH) * 8192);
            tg[(J + MID_DEPTH) % NB] = buf_ld2(r.g, v * 8, (J + MID_DEPTH) * 4096);
        }
        const cpx T = twmul<J, -1>(J < 8 ? tlo : thi);
        cpx xk2, xp2;
        mid_slot(d[O + J], d[O + 16 + 15 - J], T, tw[J % NB], tg[J % NB], xk2, xp2, chi);
        if constexpr (J == 0) L.xlow[v] = xk2;               // 2 X_k, k = v < 512
        if constexpr (WIDE) {
            if constexpr (J >= 1 && 1024 * J < NS_MAX)
                buf_st2(r.xw, xk2, (v + 1024 * J - NLOW_MAX) * 8);
            if constexpr (1024 * (15 - J) + 512 < NS_MAX)
                buf_st2(r.xw, xp2, v == 0 ? OOB_OFF : (1024 * (16 - J) - v - NLOW_MAX) * 8);
            if constexpr (O == 0 && J >= 8 && 512 + 1024 * (J - 8) < NS_MAX)
                buf_st2(r.xw, xk2, v == 0 ? 1024 * (J - 8) * 8 : OOB_OFF);
        }
        mid_unrolled<O, J + 1, NB, WIDE>(d, r, v, L, tlo, thi, tw, tg, chi);
    }
}
template <int O, bool WIDE>
__device__ __forceinline__ cpx middle_slots(cpx (&d)[NV], const MidRsrc& r, int v, FusedLds& L,
                                            cpx tlo, cpx thi, cpx chi) {
    constexpr int NB = MID_DEPTH + 1;
    float4 tw[NB];
    cpx tg[NB];
#pragma unroll
    for (int j = 0; j < MID_DEPTH; ++j) {
        tw[j] = buf_ld4(r.w, v * 16, j * 8192);
        tg[j] = buf_ld2(r.g, v * 8, j * 4096);
    }
    mid_unrolled<O, 0, NB, WIDE>(d, r, v, L, tlo, thi, tw, tg, chi);
    return chi;
}

// Same, with block B (d[O+16 .. O+32)) parked in LDS while the slots run: used when a
// workgroup has only 128 VGPRs (one virtual thread per thread, two workgroups per CU).
// xs: this thread's 16-entry column of the exchange buffer (element j at xs[j * FT]).
template <int O, int J>
__device__ __forceinline__ void mid_staged_unrolled(cpx (&d)[NV], const MidRsrc& r, int v,
                                                    FusedLds& L, cpx* xs, cpx tlo, cpx thi,
                                                    float4 (&tw)[2], cpx (&tg)[2], cpx& zn,
                                                    cpx& chi) {
    if constexpr (J < 16) {
        cpx zp = zn;
        if constexpr (J + 1 < 16) {
            tw[(J + 1) & 1] = buf_ld4(r.w, v * 16, (J + 1) * 8192);
            tg[(J + 1) & 1] = buf_ld2(r.g, v * 8, (J + 1) * 4096);
            zn = xs[(15 - (J + 1)) * FT];
        }
        const cpx T = twmul<J, -1>(J < 8 ? tlo : thi);
        cpx xk2, xp2;
        mid_slot(d[O + J], zp, T, tw[J & 1], tg[J & 1], xk2, xp2, chi);
        xs[(15 - J) * FT] = zp;
        if constexpr (J == 0) L.xlow[v] = xk2;               // 2 X_k, k = v < 512
        mid_staged_unrolled<O, J + 1>(d, r, v, L, xs, tlo, thi, tw, tg, zn, chi);
    }
}
template <int O>
__device__ __forceinline__ cpx middle_slots_staged(cpx (&d)[NV], const MidRsrc& r, int v,
                                                   FusedLds& L, cpx* xs, cpx tlo, cpx thi,
                                                   cpx chi) {
    float4 tw[2];
    cpx tg[2];
    tw[0] = buf_ld4(r.w, v * 16, 0);
    tg[0] = buf_ld2(r.g, v * 8, 0);
    cpx zn = xs[15 * FT];
    mid_staged_unrolled<O, 0>(d, r, v, L, xs, tlo, thi, tw, tg, zn, chi);
    return chi;
}

// Inter-stage twiddles w_M^{n' k1} (F1: multiply, I1: multiply by the conjugate), built
// per virtual thread from ten anchors  B_b = w^{n' b} (b = 1..7),  A_a = w^{8 n' a}
// (a = 1..3):  w^{n' (8a + b)} = A_a B_b.  80 bytes per virtual thread from L2 instead
// of 248, one load latency instead of four, and 21 extra complex products (42 packed
// instructions).  t1a[r][n'] = (anchor 2r, anchor 2r+1), anchors ordered B1..B7, A1..A3.
struct T1Anch {
    float4 q[5];
};
__device__ __forceinline__ T1Anch t1_load(__amdgpu_buffer_rsrc_t t1a, int vt) {
    T1Anch r;
#pragma unroll
    for (int c = 0; c < 5; ++c) r.q[c] = buf_ld4(t1a, vt * 16, c * 8192);
    return r;
}
template <bool CONJ, int O>
__device__ __forceinline__ void t1_apply(cpx (&d)[NV], const T1Anch& an) {
#ifdef ABL_NOFFT
    d[O] = d[O] + lo2(an.q[0]) + hi2(an.q[4]);      // keep the anchor loads alive
    return;
#endif
    cpx B[8], A[4];
#pragma unroll
    for (int i = 1; i < 8; ++i) B[i] = (i & 1) ? lo2(an.q[(i - 1) >> 1]) : hi2(an.q[(i - 1) >> 1]);
    A[1] = hi2(an.q[3]);
    A[2] = lo2(an.q[4]);
    A[3] = hi2(an.q[4]);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int k1 = 8 * a + b;
            if (k1 == 0) continue;
            const cpx w = (a == 0) ? B[b] : (b == 0) ? A[a] : cmul(A[a], B[b]);
            d[O + k1] = CONJ ? cmulc(d[O + k1], w) : cmul(d[O + k1], w);
        }
}

// Roles and LDS index maps of one virtual thread.
struct Roles {
    int vt, n3u, k1u, kB;
    __device__ __forceinline__ explicit Roles(int v) {
        vt = v;
        k1u = v >> 4; n3u = v & 15;          // F2 / I2 role (k1-major)
        kB = partner_block(v);               // F3 / I3 role
    }
    __device__ __forceinline__ int e1w(int k1) const { return k1 * LD1 + vt; }
    __device__ __forceinline__ int e1r(int n2) const { return k1u * LD1 + 16 * n2 + n3u; }
    __device__ __forceinline__ int e2w(int k2) const { return (k1u + 32 * k2) * LD2 + n3u; }
    __device__ __forceinline__ int e2r(int j) const { return (j < 16 ? vt : kB) * LD2 + (j & 15); }
};

// ------------------------------------------------------------------ the kernel
// FEAT bit 0: plan has searches that are not full-range (scan the LDS lag dump) or that
//             interpolate (neighbour amplitudes of the winner)
// FEAT bit 1: plan has time-domain windows
// FEAT bit 2: channel algebra on load (sum_j weight_j * channel_j)
// FEAT bit 3: WIDE -- a lowchi2 cut-off or psd_amp band reaches beyond NLOW_MAX bins: bins
//             512 .. nstash-1 of 2 X_k go through this workgroup's stash `xwide`
// MULTI: several filter slots (template_tag x csd_tag) share the forward transform of a
//        trace: the spectrum (state after F3) is parked in a per-workgroup scratch area
//        (128 KiB, L2 / MALL resident) and every slot runs middle -> inverse -> tail on
//        it.  Slots come from `slots[0 .. nslots)`; sd / tabs then only carry the shared
//        twiddle tables.  Not MULTI: the one slot is (sd, tabs), as kernel arguments.
template <int FEAT, bool MULTI>
__global__ __launch_bounds__(BLOCK_THREADS, (BLOCK_THREADS / 256) * WG_PER_CU) void k_fused(OfxPlanDev pd, OfxSlotDev sd, FusedTabs tabs,
                                                 const float* __restrict__ traces,
                                                 const uint8_t* __restrict__ valid,
                                                 long long n_traces, float* __restrict__ out,
                                                 const FusedSlotArg* __restrict__ slots,
                                                 int nslots, float2* __restrict__ spec,
                                                 float2* __restrict__ xwide, int nstash) {
    constexpr bool WIDE = (FEAT & 8) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // half: which of the workgroup's traces this wave works on (wave-uniform); wg: index of
    // that trace slot in the grid (scratch areas, stamps); tid / wave: index within the half
    const int half = PP ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FT)) : 0;
    const int wg = (int)blockIdx.x * NHALF + half;
    FusedShared& SH = *reinterpret_cast<FusedShared*>(smem_raw);
    FusedLds& L = reinterpret_cast<FusedLds*>(smem_raw + sizeof(FusedShared))[half];
    const int tid = PP ? (int)(threadIdx.x & (FT - 1)) : (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    // slot barrier of the ping-pong schedule inside a register-arithmetic phase (see exchange)
#define PPB()                                  \
    do {                                       \
        if constexpr (PP) __syncthreads();     \
    } while (0)
    const int pre = pd.pre;
    const __amdgpu_buffer_rsrc_t t1q = make_rsrc(tabs.t1, 5 * 512 * 16);
    const __amdgpu_buffer_rsrc_t rtb = make_rsrc(tabs.tbase, 512 * 8);

    for (int i = tid; i < 512; i += FT) SH.t2[i] = mk(tabs.t2[i].x, tabs.t2[i].y);

    const size_t ev_stride = (size_t)pd.n_channels * FN;
    cpx d[NV];

    // LDS exchange of the NV values of a thread.  widx / ridx map (role h, value j) to
    // an element index of the full layout; wpass / rpass give the half (0 / 1) that
    // element belongs to, hb the elements per half.
    //
    // PP: the buffer is shared by the two halves of the workgroup, which run the same program
    // two barriers apart.  An exchange is  [B] writes [B] reads [B]  and every arithmetic phase
    // between two exchanges holds exactly one barrier (PPB) -- so while this half writes / reads,
    // the partner is in the first / second part of an arithmetic phase, and the other way
    // round: the buffer has one user at a time and LDS time overlaps VALU time by construction.
    // Rule for every use of SH.xb: at most two barrier-delimited segments long, and at least
    // one barrier between the end of one use and the start of the next.  Every barrier is
    // executed by all eight waves the same number of times: no barrier sits under a condition
    // that depends on the half or on the data.
    auto exchange = [&](auto widx, auto wpass, auto ridx, auto rpass, int hb) {
#ifdef ABL_NOEXCH
        return;
#endif
        cpx* xc = reinterpret_cast<cpx*>(SH.xb);
#if OFX_XPRIO          // the latency chain of an exchange outranks the partner's arithmetic
        __builtin_amdgcn_s_setprio(OFX_XPRIO);
#endif
        if constexpr (!SPLIT_EXCHANGE) {
            __syncthreads();                   // earlier readers of xb are done
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int j = 0; j < 32; ++j) xc[widx(h, j)] = d[32 * h + j];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int j = 0; j < 32; ++j) d[32 * h + j] = xc[ridx(h, j)];
            PPB();                             // reads done: the partner may write
        } else {
            cpx nd[NV];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                __syncthreads();               // earlier readers of xb are done
#pragma unroll
                for (int h = 0; h < VT; ++h)
#pragma unroll
                    for (int j = 0; j < 32; ++j)
                        if (wpass(h, j) == p) xc[widx(h, j) - p * hb] = d[32 * h + j];
                __syncthreads();
#pragma unroll
                for (int h = 0; h < VT; ++h)
#pragma unroll
                    for (int j = 0; j < 32; ++j)
                        if (rpass(h, j) == p) nd[32 * h + j] = xc[ridx(h, j) - p * hb];
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) d[j] = nd[j];
        }
#if OFX_XPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // D1 exchange for one virtual thread per hardware thread and a half-size buffer:
    // "block-diagonal" passes.  Pass p moves the elements (k1, n2) with
    // (k1 >> 4) ^ (n2 >> 4) == p, so that in every pass every thread writes 16 values
    // and reads 16 values (at most 32 complex live).  Writer side of E1: thread n'
    // (n2 = n' >> 4, so n2 >> 4 = hw = wave >> 2) holds k1 = 0..31; reader side:
    // thread (k1u, n3u) with k1u >> 4 = hw holds n2 = 0..31.  E4 swaps the roles.
    // hw is wave-uniform: the two register halves are handled by two code paths behind a
    // scalar branch (the empty asm keeps the compiler from turning them into selects).
    constexpr bool DIAG_D1 = (VT == 1) && SPLIT_EXCHANGE;
    auto exchange_d1 = [&](const Roles& R, bool e4) {
        cpx* xc = reinterpret_cast<cpx*>(SH.xb);
        const int hw = __builtin_amdgcn_readfirstlane(R.vt >> 8);
        const int rowb = (R.k1u & 15) * LD1 + R.n3u;       // (k1u, n3u) side, + 16 n2
        cpx lo[16], hi[16];                                 // new values, halves 0 / 1
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool up = (p ^ hw) != 0;                 // upper half of my values moves
            __syncthreads();
            if (!e4) {                                     // E1: n' side writes
                if (!up) {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) xc[j * LD1 + R.vt] = d[j];
                } else {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) xc[j * LD1 + R.vt] = d[16 + j];
                }
            } else {                                       // E4: (k1u, n3u) side writes
                if (!up) {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) xc[rowb + 16 * j] = d[j];
                } else {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) xc[rowb + 16 * (16 + j)] = d[16 + j];
                }
            }
            __syncthreads();
            if (!e4) {
                if (!up) {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) lo[j] = xc[rowb + 16 * j];
                } else {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) hi[j] = xc[rowb + 16 * (16 + j)];
                }
            } else {
                if (!up) {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) lo[j] = xc[j * LD1 + R.vt];
                } else {
                    asm volatile("");
#pragma unroll
                    for (int j = 0; j < 16; ++j) hi[j] = xc[j * LD1 + R.vt];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d[j] = lo[j];
            d[16 + j] = hi[j];
        }
    };

    // Trace load into the register file (virtual thread vt reads z[512 n1 + vt]).
    // FEAT & 4 (channel algebra): the request only carries the first term, unscaled, so that it
    // stays a plain asynchronous prefetch; the weight and the other terms are applied by
    // combine_terms when the trace is taken up (with the FMA next to the loads the "prefetch" of a
    // 4-channel plan waited for every load in turn: 72 k cycles per trace in the timeline of
    // BASELINE configs[3]).  The arithmetic is unchanged: w0 s0, then fma(w_c, s_c, .) in order.
    auto load_trace = [&](long long bb) __attribute__((always_inline)) {
        int tl = ofx_fresh_tid(wave_base);
        const float* e = traces + (size_t)bb * ev_stride;
        const __amdgpu_buffer_rsrc_t rz =
            make_rsrc(e + ((FEAT & 4) ? (size_t)pd.chan[0] * FN : 0), FN * 4);
#pragma unroll
        for (int h = 0; h < VT; ++h)
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1)
                d[32 * h + n1] = buf_ld2(rz, (tl + FT * h) * 8, n1 * 4096);
    };
    auto combine_terms = [&](long long bb) __attribute__((always_inline)) {
        if constexpr (FEAT & 4) {
            if (pd.n_terms == 1 && pd.weight[0] == 1.0f) return;         // uniform: a plain select
            int tl = ofx_fresh_tid(wave_base);
            const float* e = traces + (size_t)bb * ev_stride;
            const float w0 = pd.weight[0];
#pragma unroll
            for (int j = 0; j < NV; ++j) d[j] = d[j] * mk(w0, w0);
            for (int c = 1; c < pd.n_terms; ++c) {
                const __amdgpu_buffer_rsrc_t rz = make_rsrc(e + (size_t)pd.chan[c] * FN, FN * 4);
                const float wgt = pd.weight[c];
#pragma unroll
                for (int h = 0; h < VT; ++h)
#pragma unroll
                    for (int n1 = 0; n1 < 32; ++n1) {
                        const cpx s = buf_ld2(rz, (tl + FT * h) * 8, n1 * 4096);
                        d[32 * h + n1] = pfma(mk(wgt, wgt), s, d[32 * h + n1]);
                    }
            }
        }
    };

    // MULTI: the spectrum of the current trace, [value j][thread] in this workgroup's area.
    const __amdgpu_buffer_rsrc_t rspec =
        make_rsrc(spec + (MULTI ? (size_t)wg * NV * FT : 0), NV * FT * 8);
    auto store_spec = [&]() __attribute__((always_inline)) {
        int tl = ofx_fresh_tid(wave_base);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u32x2 v;
            v.x = __float_as_uint(d[j].x);
            v.y = __float_as_uint(d[j].y);
            __builtin_amdgcn_raw_buffer_store_b64(v, rspec, tl * 8, j * FT * 8, 0);
        }
    };
    auto load_spec = [&]() __attribute__((always_inline)) {
        int tl = ofx_fresh_tid(wave_base);
#pragma unroll
        for (int j = 0; j < NV; ++j) d[j] = buf_ld2(rspec, tl * 8, j * FT * 8);
    };

    // Software-pipelined trace load: the next trace is requested as soon as the
    // registers of the current one are dead (after the arg-max), so that its HBM
    // latency hides under the rest of the tail; `have` = d already holds trace b.
    bool have = false;
    [[maybe_unused]] int tdpar = 0;          // parity of the trace count: which half of L.tdred is written
#ifdef OFX_STAMPS
    int stamp_it = 0;
    unsigned long long* stamp_base;
    {
        const size_t off = (((size_t)wg * OFX_STAMP_TRACES) * NWAVE +
                            (size_t)__builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & (NWAVE - 1))) * 16;
        // (the stamps live behind the stash area of the WIDE variants in the same buffer)
        const unsigned long long a = reinterpret_cast<unsigned long long>(
            xwide + (size_t)gridDim.x * NHALF * (NS_MAX - NLOW_MAX)) + off * 8;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        stamp_base = reinterpret_cast<unsigned long long*>(((unsigned long long)hi << 32) | lo);
    }
#endif
    // PP: the second half starts two barriers late (and the first ends two barriers late)
    if constexpr (PP) {
        if (half == 1) {
            __syncthreads();
            __syncthreads();
        }
    }
    // Trace slots are handed out in groups of NHALF; a half whose slot lies beyond the batch
    // (odd batch) works on the last trace again and writes nothing, and an event flagged
    // invalid runs through the whole pipeline and gets the sentinel row at the end: every wave
    // executes the same sequence of barriers whatever the data.
    const long long stride = (long long)gridDim.x * NHALF;
    for (long long b0 = (long long)blockIdx.x * NHALF; b0 < n_traces; b0 += stride) {
        const bool dummy = (b0 + half >= n_traces);
        const long long b = dummy ? n_traces - 1 : b0 + half;
        float* row = out + (size_t)b * pd.row;
        const bool skip = dummy || (valid && !valid[b]);     // wave-uniform
        // (the sentinel row goes out first -- nothing else is written for such an event; with
        // this loop after the pipeline, next to the prefetched registers of the following
        // trace, the allocator spilled the whole data array)
        if (skip && !dummy)
            for (int j = tid; j < pd.row; j += FT) row[j] = OFX_SENTINEL;
        if constexpr (!PP) {       // independent workgroups: an invalid event is simply skipped
            if (skip) {
                have = false;
                continue;
            }
        }
        // Roles are re-derived from an opaque copy of tid every trace so that the LDS
        // address arithmetic stays next to its use (LICM would hoist and spill it).
        int tl = ofx_fresh_tid(wave_base);
        const Roles R0(tl), R1(tl + FT);
        auto RR = [&](int h) -> const Roles& { return h == 0 ? R0 : R1; };

        STAMP(0);                                // loop overhead / previous tail remainder
        if (!have) load_trace(b);               // cold start / after an invalid event
        combine_terms(b);

        // ------------------------------------------------ time-domain windows
        // Sample index of d[32 h + n1].{x,y} is 1024 n1 + 2 vt + {0,1}: a window [lo, hi)
        // covers whole rows n1 (uniform test, unmasked adds / max3 / min3), at most two
        // partial rows (masked), and rows outside it are skipped.
        if constexpr (FEAT & 2) {
            tdpar ^= 1;
            // end points of the slice owned by thread w (trapezoid correction): requested
            // first, consumed after the reductions
#include "ofx_fused_td_endpoints.inc"
            for (int w = 0; w < pd.n_tdwin; ++w) {
                const int lo = pd.tdw[w].lo, hi = pd.tdw[w].hi;
                float s = 0.0f, sq = 0.0f, mx = -INFINITY, mn = INFINITY;
                cpx s2 = mk(0.0f, 0.0f), sq2 = mk(0.0f, 0.0f);            // full rows, packed
                // (row classification done by the host per launch: two bit tests per row instead of
                // four compares)
                const unsigned fullm = pd.tdw[w].full, anym = fullm | pd.tdw[w].edge;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    const int r0 = 1024 * n1;
                    if (!((anym >> n1) & 1u)) continue;                   // uniform: outside
                    if ((fullm >> n1) & 1u) {                             // uniform: full row
#pragma unroll
                        for (int h = 0; h < VT; ++h) {
                            // (dependent forms only: anything that is a function of d
                            // alone would be hoisted out of the window loop and spilled)
                            const cpx v = d[32 * h + n1];
                            s2 = s2 + v;
                            sq2 = pfma(v, v, sq2);
                            mx = max3f(mx, v.x, v.y);
                            mn = min3f(mn, v.x, v.y);
                        }
                    } else {                                              // edge row
                        // (the sample index against the window, as the thread's offset in the row
                        // against SCALAR bounds: with `n = r0 + 2 (tl + FT h)` per row the compiler
                        // hoisted all 128 values of n, n + 1 out of the window loop -- 470 B of
                        // scratch per lane in every kernel with windows, 80 scratch reloads per
                        // trace each behind an s_waitcnt vmcnt(0))
                        const int lo_r = lo - r0, hi_r = hi - r0;         // uniform
#pragma unroll
                        for (int h = 0; h < VT; ++h) {
                            const int c = 2 * (tl + FT * h);              // offset in the row
                            const bool in0 = (c >= lo_r) && (c < hi_r);
                            const bool in1 = (c >= lo_r - 1) && (c < hi_r - 1);
                            const cpx v = d[32 * h + n1];
                            const float y0 = in0 ? v.x : 0.0f, y1 = in1 ? v.y : 0.0f;
                            s = (s + y0) + y1;
                            sq = fmaf(y0, y0, fmaf(y1, y1, sq));
                            mx = max3f(mx, in0 ? v.x : -INFINITY, in1 ? v.y : -INFINITY);
                            mn = min3f(mn, in0 ? v.x : INFINITY, in1 ? v.y : INFINITY);
                        }
                    }
                }
                s = ofx_wave_sum(s + (s2.x + s2.y));
                sq = ofx_wave_sum(sq + (sq2.x + sq2.y));
                mx = ofx_wave_max(mx);
                mn = ofx_wave_min(mn);
                if (lane == 0) {
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
        {   // diagnostic: stream the traces and do nothing else
            cpx acc = mk(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NV; ++j) acc += d[j];
            if (acc.x + acc.y == 1.2345f) row[0] = acc.x;
            have = false;
            continue;
        }
#endif
        STAMP(1);                                // wait for the trace + TD windows
        // ---------------------------------------------------------------- F1
        if constexpr (VT == 2) {
            const T1Anch g0 = t1_load(t1q, tl);
            const T1Anch g1 = t1_load(t1q, tl + FT);
            __builtin_amdgcn_sched_barrier(0);         // keep the requests ahead of the DFTs
            dft<32, -1, NV, 0>(d);
            dft<32, -1, NV, 32 * (VT - 1)>(d);
            t1_apply<false, 0>(d, g0);
            t1_apply<false, 32 * (VT - 1)>(d, g1);
        } else {
            dft<32, -1, NV, 0>(d);
            __builtin_amdgcn_sched_barrier(0);         // 128-VGPR build: no early requests
            t1_apply<false, 0>(d, t1_load(t1q, tl));
        }
        STAMP(2);                                // F1
        if constexpr (DIAG_D1) {
            exchange_d1(R0, false);
        } else {
            exchange([&](int h, int j) { return RR(h).e1w(j); }, [](int, int j) { return j >> 4; },
                     [&](int h, int j) { return RR(h).e1r(j); }, [](int h, int) { return h; },
                     HB1);
        }
        STAMP(3);                                // E1
        // ---------------------------------------------------------------- F2
        dft<32, -1, NV, 0>(d);
        PPB();
        if constexpr (VT == 2) dft<32, -1, NV, 32 * (VT - 1)>(d);
#pragma unroll
        for (int h = 0; h < VT; ++h)
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2)
                d[32 * h + k2] = cmul(d[32 * h + k2], SH.t2[k2 * 16 + RR(h).n3u]);
        STAMP(4);                                // F2
        exchange([&](int h, int j) { return RR(h).e2w(j); }, [](int, int j) { return j >> 4; },
                 [&](int h, int j) { return RR(h).e2r(j); }, [](int, int j) { return j >> 4; },
                 HB2);
        STAMP(5);                                // E2
#if OFX_MPRIO          // the middle step is fed by filter-table loads
        __builtin_amdgcn_s_setprio(OFX_MPRIO);
#endif
        // Everything from here to the output row depends on the filter slot.  Not MULTI:
        // one pass on the kernel-argument slot (the loop and the selections fold away).
#define SDX (MULTI ? slots[slot_i].sd : sd)
#define TBX (MULTI ? slots[slot_i].tabs : tabs)
        if constexpr (MULTI) {
            // F3 for every block, park the spectrum, then one pass per slot
            dft<16, -1, NV, 0>(d);
            dft<16, -1, NV, 16>(d);
            if constexpr (VT == 2) {
                dft<16, -1, NV, 32>(d);
                dft<16, -1, NV, 48>(d);
            }
            store_spec();
        }
        const int slot_n = MULTI ? nslots : 1;
        for (int slot_i = 0; slot_i < slot_n; ++slot_i) {
        // ------------------------------------------- F3, middle, I3 (registers)
        const MidRsrc rmid = {make_rsrc(TBX.midW, 16 * 512 * 16),
                              make_rsrc(TBX.midG, 16 * 512 * 8),
                              make_rsrc(xwide + (WIDE ? (size_t)wg * (NS_MAX - NLOW_MAX) : 0),
                                        (WIDE && slot_i == 0) ? (nstash - NLOW_MAX) * 8 : 0)};
        cpx chi2v = mk(0.0f, 0.0f);
        constexpr bool STAGE_B = (VT == 1) && SPLIT_EXCHANGE;
        if constexpr (STAGE_B) {
            if constexpr (!MULTI) {
                dft<16, -1, NV, 0>(d);
                dft<16, -1, NV, 16>(d);
            }
            const cpx a8 = d[8];
            if (wave == 0) perm_in<0>(d, tl == 0, L.perm);
            cpx* xs = reinterpret_cast<cpx*>(SH.xb) + tl;
            const cpx tb = buf_ld2(rtb, tl * 8, 0);
            const cpx tbh = (tl == 0) ? mk(tabs.tb0hi.x, tabs.tb0hi.y) : tb;   // (the kernel argument: see Tabs25::tb0hi)
            __syncthreads();                   // every E2 read is done: the buffer is free
#pragma unroll
            for (int j = 0; j < 16; ++j) xs[j * FT] = d[16 + j];
            chi2v = middle_slots_staged<0>(d, rmid, tl, L, xs, tb, tbh, chi2v);
#pragma unroll
            for (int j = 0; j < 16; ++j) d[16 + j] = xs[j * FT];
            if (wave == 0) chi2v = perm_out<0>(d, tl == 0, a8, TBX, chi2v, L.perm);
            dft<16, +1, NV, 0>(d);
            dft<16, +1, NV, 16>(d);
        } else {
            if constexpr (!MULTI) {
                dft<16, -1, NV, 0>(d);
                dft<16, -1, NV, 16>(d);
            }
            const cpx a8 = d[8];
            if (wave == 0) perm_in<0>(d, tl == 0, L.perm);
            const cpx tb = buf_ld2(rtb, tl * 8, 0);
            const cpx tbh = (tl == 0) ? mk(tabs.tb0hi.x, tabs.tb0hi.y) : tb;   // (the kernel argument: see Tabs25::tb0hi)
            chi2v = middle_slots<0, WIDE>(d, rmid, tl, L, tb, tbh, chi2v);
            if (wave == 0) chi2v = perm_out<0>(d, tl == 0, a8, TBX, chi2v, L.perm);
            dft<16, +1, NV, 0>(d);
            dft<16, +1, NV, 16>(d);
        }
        PPB();
        if constexpr (VT == 2) {
            constexpr int O = 32 * (VT - 1);
            if constexpr (!MULTI) {
                dft<16, -1, NV, O>(d);
                dft<16, -1, NV, O + 16>(d);
            }
            const cpx tb = buf_ld2(rtb, (tl + FT) * 8, 0);
            chi2v = middle_slots<O, WIDE>(d, rmid, tl + FT, L, tb, tb, chi2v);
            dft<16, +1, NV, O>(d);
            dft<16, +1, NV, O + 16>(d);
        }
        const float chi0p = chi2v.x + chi2v.y;
        // (the wave's share of chi2_0 goes to LDS here: carried to the tail in a register it was spilled
        // across the inverse transform and reloaded behind an s_waitcnt vmcnt(0))
        {
            const float wchi = ofx_wave_sum(chi0p);
            if (lane == 0) L.red[1][wave_base >> 6] = wchi;     // (scalar index: no address register to keep)
        }
#if OFX_MPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(6);                                // F3 + middle + I3
        int tl2 = ofx_fresh_tid(wave_base);          // no CSE of addresses across the middle
        const Roles Q0(tl2), Q1(tl2 + FT);
        auto QQ = [&](int h) -> const Roles& { return h == 0 ? Q0 : Q1; };
        exchange([&](int h, int j) { return QQ(h).e2r(j); }, [](int, int j) { return j >> 4; },
                 [&](int h, int j) { return QQ(h).e2w(j); }, [](int, int j) { return j >> 4; },
                 HB2);
        STAMP(7);                                // E3
        // ---------------------------------------------------------------- I2
#pragma unroll
        for (int h = 0; h < VT; ++h)
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2)
                d[32 * h + k2] = cmulc(d[32 * h + k2], SH.t2[k2 * 16 + QQ(h).n3u]);
        dft<32, +1, NV, 0>(d);
        PPB();
        if constexpr (VT == 2) dft<32, +1, NV, 32 * (VT - 1)>(d);
        STAMP(8);                                // I2
        {
            T1Anch g0, g1;
            if constexpr (!DIAG_D1) {                  // requests ahead of the exchange
                g0 = t1_load(t1q, tl2);
                g1 = t1_load(t1q, tl2 + FT);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DIAG_D1) {
                exchange_d1(Q0, true);
                g0 = t1_load(t1q, tl2);
                g1 = g0;
            } else {
                exchange([&](int h, int j) { return QQ(h).e1r(j); }, [](int h, int) { return h; },
                         [&](int h, int j) { return QQ(h).e1w(j); },
                         [](int, int j) { return j >> 4; }, HB1);
            }
            STAMP(9);                            // E4
            // ------------------------------------------------------------ I1
            t1_apply<true, 0>(d, g0);
            if constexpr (VT == 2) t1_apply<true, 32 * (VT - 1)>(d, g1);
        }
        dft<32, +1, NV, 0>(d);
        if constexpr (VT == 2) dft<32, +1, NV, 32 * (VT - 1)>(d);
        // d[32 h + n1] = (A(1024 n1 + 2 vt), A(1024 n1 + 2 vt + 1)),  vt = tid + FT h

#ifdef ABL_NOTAIL
        {
            float acc = chi0p;
#pragma unroll
            for (int j = 0; j < NV; ++j) acc += d[j].x + d[j].y;
            if (acc == 1.2345f) row[0] = acc;
            PPB();
            continue;
        }
#endif
        STAMP(10);                               // I1
#if OFX_TPRIO          // the tail is a latency chain as well
        __builtin_amdgcn_s_setprio(OFX_TPRIO);
#endif
        // ------------------------------------------------------------- tail
        // (thread ids of the tail come from an opaque copy: its LDS / table addresses are
        // recomputed here instead of being hoisted out of the loop and spilled)
        int tt = ofx_fresh_tid(wave_base);
        const int lane_t = tt & 63, wave_t = tt >> 6;
        const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(SDX.s, NLOW_MAX * 8);
        const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(SDX.g, NLOW_MAX * 4);
        // max of A^2 per group of 8 register pairs (kept so that the thread holding the
        // global maximum only has to search one group), then per thread
        constexpr int NG = NV / 8;
        float gm[NG];
        float mloc = 0.0f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float m = 0.0f;
#pragma unroll
            for (int j = 8 * g; j < 8 * g + 8; ++j) {
                const cpx sq = d[j] * d[j];
                m = fmaxf(m, fmaxf(sq.x, sq.y));
            }
            gm[g] = m;
            mloc = fmaxf(mloc, m);
        }
        // low-frequency chi2 tables for this thread's bins: requested now, used at the end
        constexpr int NLK = (NLOW_MAX + FT - 1) / FT;
        cpx lk_s[NLK];
        float lk_g[NLK];
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            lk_s[i] = buf_ld2(rs_s, (tt + FT * i) * 8, 0);
            lk_g[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rs_g, (tt + FT * i) * 4, 0, 0));
        }
        TSTAMP(2);                               // group maxima, table requests
        {
            const float wmax = ofx_wave_max(mloc);
            __syncthreads();
            if (lane_t == 0) L.red[0][wave_t] = wmax;
            if (tt == 0) L.bcast[0] = d[0].x;          // A(lag 0) for nodelay
            __syncthreads();
        }
        float Mstar = L.red[0][0], chi0 = L.red[1][0];
#pragma unroll
        for (int q = 1; q < NWAVE; ++q) {
            Mstar = fmaxf(Mstar, L.red[0][q]);
            chi0 += L.red[1][q];
        }
        float a_lag0 = L.bcast[0];
        // one register each from here on (the sums would otherwise be sunk to their uses
        // with every per-wave partial kept alive)
        asm volatile("" : "+v"(Mstar), "+v"(chi0), "+v"(a_lag0));
        TSTAMP(3);                               // block maximum and chi2_0

        // full-range delay fit: the thread(s) holding the maximum resolve the
        // smallest rolled index among their lags with A^2 == max
        OfxCand fullbest = ofx_cand_none();
        bool any_full = false;
#pragma unroll 1
        for (int q = 0; q < SDX.n_search; ++q) {
            const OfxSearchDev& sq = SDX.search[q];
            any_full |= (sq.kind == OFX_SEARCH_DELAY) && !sq.outside && sq.lo == 0 &&
                        sq.hi == FN;
        }
        if (any_full) {
            if (mloc == Mstar) {
                const int tb = tt;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (__builtin_amdgcn_ballot_w64(gm[g] == Mstar) == 0) continue;   // uniform
                    const int h = (8 * g) / 32;
                    const int base = 2 * (tb + FT * h) + pre;
#pragma unroll
                    for (int j = 8 * g; j < 8 * g + 8; ++j) {
                        const int n1 = j & 31;
                        const cpx v = d[j];
                        const int i0 = (base + 1024 * n1) & (FN - 1);
                        const int i1 = (base + 1024 * n1 + 1) & (FN - 1);
                        if (v.x * v.x == Mstar && i0 < fullbest.idx) {
                            fullbest.idx = i0; fullbest.amp = v.x; fullbest.key = Mstar;
                        }
                        if (v.y * v.y == Mstar && i1 < fullbest.idx) {
                            fullbest.idx = i1; fullbest.amp = v.y; fullbest.key = Mstar;
                        }
                    }
                }
            }
            fullbest = ofx_cand_block_reduce(fullbest, L.cand, tt, NWAVE);
        }

        TSTAMP(4);                               // full-range arg-max
        // windowed / outside-window fits scan the lag dump: one use of the exchange buffer,
        // two barrier-delimited segments (dump | scans of every windowed search, each wave
        // keeping its winner per search in L.wc; the waves' winners are merged by resolve)
        if constexpr (FEAT & 1) {
            if (lane_t < OFX_MAX_SEARCHES) L.wc[lane_t][wave_t] = ofx_cand_none();
            constexpr int NPASS = SPLIT_EXCHANGE ? 2 : 1;   // SPLIT: even lags, then odd
            // Narrow windows (the usual constrained fit: +-400 us is 1000 lags, one or two of the 32
            // register rows of 1024 lags): only the rows the windowed searches touch are dumped,
            // complex, in ONE pass (row n1 -> slot popcount(rowmask below n1); the host computes the
            // mask per launch).  Wider or outside-window searches keep the dump of all lags.
            const unsigned rmask = TBX.rowmask;
            constexpr int NSLOT = (SPLIT_EXCHANGE ? XBUF_ELEMS : 2 * XBUF_ELEMS) / 1024;
            if (SPLIT_EXCHANGE && !PP && __builtin_popcount(rmask) <= NSLOT) {
                cpx* xc = reinterpret_cast<cpx*>(SH.xb);
                __syncthreads();
                int slotb = 0;                                   // uniform: slot * 512
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    if (!((rmask >> n1) & 1u)) continue;         // uniform
#pragma unroll
                    for (int h = 0; h < VT; ++h) xc[slotb + tt + FT * h] = d[32 * h + n1];
                    slotb += 512;
                }
                __syncthreads();
#pragma unroll 1
                for (int q = 0; q < SDX.n_search; ++q) {
                    const OfxSearchDev& sq = SDX.search[q];
                    const bool full = !sq.outside && sq.lo == 0 && sq.hi == FN;
                    if (sq.kind != OFX_SEARCH_DELAY || full) continue;
                    OfxCand c = ofx_cand_none();
                    auto scan = [&](int i0, int i1) {
                        for (int i = i0 + tt; i < i1; i += FT) {
                            const int n = (i - pre) & (FN - 1);
                            const int n1 = n >> 10;
                            const int sl = __builtin_popcount(rmask & ((1u << n1) - 1u));
                            ofx_cand_take(c, SH.xb[sl * 1024 + (n & 1023)], i);
                        }
                    };
                    if (sq.outside) {
                        scan(0, sq.lo);
                        scan(sq.hi, FN);
                    } else {
                        scan(sq.lo, sq.hi);
                    }
                    c = ofx_cand_wave_reduce(c);
                    if (lane_t == 0) L.wc[q][wave_t] = c;
                }
            } else
            for (int e = 0; e < NPASS; ++e) {
                __syncthreads();
#pragma unroll
                for (int h = 0; h < VT; ++h)
#pragma unroll
                    for (int n1 = 0; n1 < 32; ++n1) {
                        const int m = 512 * n1 + tt + FT * h;
                        if constexpr (SPLIT_EXCHANGE) {
                            SH.xb[m] = e ? d[32 * h + n1].y : d[32 * h + n1].x;   // A(2m+e)
                        } else {
                            reinterpret_cast<cpx*>(SH.xb)[m] = d[32 * h + n1];    // A(2m), A(2m+1)
                        }
                    }
                __syncthreads();
#pragma unroll 1
                for (int q = 0; q < SDX.n_search; ++q) {
                    const OfxSearchDev& sq = SDX.search[q];
                    const bool full = !sq.outside && sq.lo == 0 && sq.hi == FN;
                    if (sq.kind != OFX_SEARCH_DELAY || full) continue;
                    OfxCand c = ofx_cand_none();
                    auto scan = [&](int i0, int i1) {
                        for (int i = i0 + tt; i < i1; i += FT) {
                            const int n = (i - pre) & (FN - 1);
                            if constexpr (SPLIT_EXCHANGE) {
                                if ((n & 1) == e) ofx_cand_take(c, SH.xb[n >> 1], i);
                            } else {
                                ofx_cand_take(c, SH.xb[n], i);
                            }
                        }
                    };
                    if (sq.outside) {
                        scan(0, sq.lo);
                        scan(sq.hi, FN);
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

        TSTAMP(5);                               // lag dump and window scans
        // psd_amp bands from the stashed 2 X_k (LDS below NLOW_MAX, the global stash above);
        // one wave per band
        const __amdgpu_buffer_rsrc_t rxw =
            make_rsrc(xwide + (WIDE ? (size_t)wg * (NS_MAX - NLOW_MAX) : 0),
                      WIDE ? (nstash - NLOW_MAX) * 8 : 0);
#include "ofx_fused_bands.inc"

        // The fit of search q (uniform): no-delay lag, full-range winner or window winner.
#include "ofx_fused_resolve.inc"
        // interpolate=True: amplitudes at the rolled bins idx -+ 1 (the last readers of d)
        if constexpr (FEAT & 1) {
#pragma unroll 1
            for (int q = 0; q < SDX.n_search; ++q) {
                const OfxSearchDev& sq = SDX.search[q];
                if (!sq.interp) continue;  // uniform
                const OfxCand best = resolve(sq, q);
                // lag n = 1024 n1 + 2 vt + e sits in thread vt % FT, register
                // 32 (vt / FT) + n1, component e: only the owning thread looks
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int n = (best.idx + (side ? 1 : -1) - pre) & (FN - 1);
                    const int vt_n = (n & 1023) >> 1;
                    if (tt == (vt_n & (FT - 1))) {
                        const int jn = 32 * (vt_n / FT) + (n >> 10);
                        cpx v = d[0];
#pragma unroll
                        for (int j = 1; j < NV; ++j) v = (j == jn) ? d[j] : v;
                        L.nb[q][side] = (n & 1) ? v.y : v.x;
                    }
                }
                __syncthreads();
                const OfxRefined ref = ofx_interpolate(L.nb[q][0], best.amp, L.nb[q][1], best.idx,
                                                       FN, SDX.norm, chi0);
                if (tt == 0) L.ref[q] = ref;
            }
            __syncthreads();
        }
        TSTAMP(6);                               // bands, interpolation
        // Per search: every thread's share of the low-frequency chi2, parked per wave in LDS.
        // The thread's bins are FT apart -- k = tt + FT i: bins 0 .. NLOW_MAX-1 from LDS, then (WIDE)
        // the first chunk of the stash, bins NLOW_MAX .. NLOW_MAX + WCH FT - 1 -- so the phase
        // exp(-2 pi i k (d + frac) / N) of one search runs along ONE chain: exp(..tt..) and the
        // uniform step exp(..FT..) from sincospif (integer part of the angle reduced exactly), every
        // further bin one complex product.  Two sincospif per search and thread instead of one per
        // bin (six at the 50 kHz cut-off of the reference's example): the terms were 13 k of the 35 k
        // cycles of a slot's tail in BASELINE configs[3] (profiles/r03_tail_timeline_config3.json).
        constexpr int WCH = 4;
        const __amdgpu_buffer_rsrc_t rw_s = make_rsrc(SDX.s, NS_MAX * 8);
        const __amdgpu_buffer_rsrc_t rw_g = make_rsrc(SDX.g, NS_MAX * 4);
        [[maybe_unused]] cpx wx[WCH], wsv[WCH];
        [[maybe_unused]] float wgv[WCH];
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                const int k = NLOW_MAX + tt + FT * i;
                wx[i] = buf_ld2(rxw, (k - NLOW_MAX) * 8, 0);
                wsv[i] = buf_ld2(rw_s, k * 8, 0);
                wgv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw_g, k * 4, 0, 0));
            }
        }
        auto phase_of = [&](int k, int dl, float frac) {      // exp(-2 pi i k (dl + frac) / N)
            const int m = (int)(((unsigned)k * (unsigned)dl) & (unsigned)(FN - 1));
            float sn, cs;
            sincospif(-2.0f * ((float)m + (float)k * frac) / (float)FN, &sn, &cs);
            return mk(cs, sn);
        };
        auto low_term = [&](int k, cpx ph, float amp, float vx, float vy, cpx S, float g) {
            const float pr = ph.x * S.x - ph.y * S.y;         // ph S
            const float pi = ph.x * S.y + ph.y * S.x;
            const float rr = vx - amp * pr;
            const float ri = vy - amp * pi;
            return ((k == 0) ? 1.0f : 2.0f) * g * (rr * rr + ri * ri);
        };
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
            cpx ph = phase_of(tt, dl, ref.frac);
            const cpx step = phase_of(FT, dl, ref.frac);      // uniform
            float low = 0.0f;
#pragma unroll
            for (int i = 0; i < NLK; ++i) {
                const int k = tt + FT * i;
                if (k < sq.nlow) {
                    const cpx x2 = L.xlow[k];
                    low += low_term(k, ph, ref.amp, 0.5f * x2.x, 0.5f * x2.y, lk_s[i], lk_g[i]);
                }
                ph = cmul(ph, step);
            }
            if constexpr (WIDE) {
                if (sq.nlow > NLOW_MAX) {                             // uniform
#pragma unroll
                    for (int i = 0; i < WCH; ++i) {
                        const int k = NLOW_MAX + tt + FT * i;
                        if (k < sq.nlow) {
                            // (entries of the partner blocks are stored conjugated, see MidRsrc)
                            const float sgn = ((k & 1023) > 512) ? -0.5f : 0.5f;
                            low += low_term(k, ph, ref.amp, 0.5f * wx[i].x, sgn * wx[i].y, wsv[i], wgv[i]);
                        }
                        ph = cmul(ph, step);
                    }
                }
            }
            low = ofx_wave_sum(low);
            if (lane_t == 0) L.lowp[q][wave_t] = low;
            if (tt == 0) L.fin[q] = best;
        }
        TSTAMP(7);                               // low-frequency chi2: LDS bins and the first stash chunk
        // ... WIDE, cut-offs beyond the first chunk (> 1536 bins: 58 kHz at 1.25 MHz): the further
        // chunks of WCH bins per thread, every search with a cut-off beyond the chunk's first bin
        // adds its share (one sincospif per bin here)
        if constexpr (WIDE) {
            int nmax = 0;
#pragma unroll 1
            for (int q = 0; q < SDX.n_search; ++q) nmax = max(nmax, SDX.search[q].nlow);
#pragma unroll 1
            for (int k0 = NLOW_MAX + WCH * FT; k0 < nmax; k0 += WCH * FT) {
                cpx cx[WCH], csv[WCH];
                float cg[WCH];
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    const int k = k0 + tt + FT * i;
                    cx[i] = buf_ld2(rxw, (k - NLOW_MAX) * 8, 0);
                    csv[i] = buf_ld2(rw_s, k * 8, 0);
                    cg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw_g, k * 4, 0, 0));
                }
#pragma unroll 1
                for (int q = 0; q < SDX.n_search; ++q) {
                    const OfxSearchDev& sq = SDX.search[q];
                    if (sq.nlow <= k0) continue;                         // uniform
                    const OfxCand best = resolve(sq, q);
                    const int dl = best.idx - pre;
                    float amp = best.amp, frac = 0.0f;
                    if constexpr (FEAT & 1)
                        if (sq.interp) {
                            amp = L.ref[q].amp;
                            frac = L.ref[q].frac;
                        }
                    float low = 0.0f;
#pragma unroll
                    for (int i = 0; i < WCH; ++i) {
                        const int k = k0 + tt + FT * i;
                        if (k < sq.nlow) {
                            const float sgn = ((k & 1023) > 512) ? -0.5f : 0.5f;
                            low += ofx_lowchi2_term(k, FN, dl, amp,
                                                    make_float2(0.5f * cx[i].x, sgn * cx[i].y),
                                                    make_float2(csv[i].x, csv[i].y), cg[i], frac);
                        }
                    }
                    low = ofx_wave_sum(low);
                    if (lane_t == 0) L.lowp[q][wave_t] += low;           // same lane wrote it above
                }
            }
        }
        STAMP(11);                               // tail A: max, reductions, arg-max, lowchi2 terms
        // The row goes out first (search q by thread q), then the next trace is requested:
        // with the loads issued first, the writer's temporaries shared registers with the load
        // destinations and the compiler made it wait for all but one of the 64 loads
        // (s_waitcnt vmcnt(1)) -- the whole HBM latency sat in front of the row write, and
        // behind it the first barrier of the next trace held the other three waves
        // (profiles/r02_phase_timeline_before.json: 9.3 k cycles in "tailB").
        __syncthreads();
#include "ofx_fused_row_write.inc"
        TSTAMP(8);                               // barrier, row write
        // d and every table value are dead: request the next trace; its HBM latency hides
        // under the loop overhead and the first stages of the other workgroup
        if (MULTI && slot_i + 1 < slot_n) {
            load_spec();                         // the spectrum again, for the next slot
        } else {
            const long long bn = b0 + stride + half;
            have = bn < n_traces;
            if (have) load_trace(bn);
        }
        }
#undef SDX
#undef TBX
#if OFX_TPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(12);                               // tail B: lowchi2 + row write
    }
    if constexpr (PP) {
        if (half == 0) {
            __syncthreads();
            __syncthreads();
        }
    }
#undef PPB
#ifdef OFX_STAMPS
    asm volatile("s_dcache_wb");
#endif
}


// ------------------------------------------------------------------ the transform on its own
// Batched complex transform of rows of FM = 16384 points, natural order in and out, unnormalised:
// the register-resident stages of k_fused (F1 E1 F2 E2 F3, or I3 E3 I2 E4 I1) without the middle
// step and the tail.  The N x M engine runs its transforms of 32768-sample traces on it instead
// of rocFFT (ofx_nxm.hip; the ROCFFT engine keeps rocFFT: the tests use it as the independent
// cross-check of k_fused).
template <bool FWD>
__global__ __launch_bounds__(FT, WG_PER_CU) void k_fft32(const float2* __restrict__ t1,
                                                         const float2* __restrict__ t2tab,
                                                         const float2* __restrict__ in,
                                                         float2* __restrict__ out, long long rows) {
    static_assert(VT == 2 && SPLIT_EXCHANGE && !PP, "product layout");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    FusedShared& SH = *reinterpret_cast<FusedShared*>(smem_raw);
    const int tid = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t t1q = make_rsrc(t1, 5 * 512 * 16);
    for (int i = tid; i < 512; i += FT) SH.t2[i] = mk(t2tab[i].x, t2tab[i].y);
    cpx d[NV];
    auto exchange = [&](auto widx, auto wpass, auto ridx, auto rpass, int hb) {
        cpx* xc = reinterpret_cast<cpx*>(SH.xb);
        cpx nd[NV];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (wpass(h, j) == p) xc[widx(h, j) - p * hb] = d[32 * h + j];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (rpass(h, j) == p) nd[32 * h + j] = xc[ridx(h, j) - p * hb];
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) d[j] = nd[j];
    };
    for (long long b = (long long)blockIdx.x; b < rows; b += (long long)gridDim.x) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const Roles R0(tl), R1(tl + FT);
        auto RR = [&](int h) -> const Roles& { return h == 0 ? R0 : R1; };
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)b * FM, FM * 8);
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(out + (size_t)b * FM, FM * 8);
        auto st2 = [&](cpx v, int idx) {
            u32x2 u;
            u.x = __float_as_uint(v.x);
            u.y = __float_as_uint(v.y);
            __builtin_amdgcn_raw_buffer_store_b64(u, rout, idx * 8, 0, 0);
        };
        if constexpr (FWD) {
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1)
                    d[32 * h + n1] = buf_ld2(rin, (tl + FT * h) * 8, n1 * 4096);
            const T1Anch g0 = t1_load(t1q, tl);
            const T1Anch g1 = t1_load(t1q, tl + FT);
            dft<32, -1, NV, 0>(d);
            dft<32, -1, NV, 32>(d);
            t1_apply<false, 0>(d, g0);
            t1_apply<false, 32>(d, g1);
            exchange([&](int h, int j) { return RR(h).e1w(j); }, [](int, int j) { return j >> 4; },
                     [&](int h, int j) { return RR(h).e1r(j); }, [](int h, int) { return h; }, HB1);
            dft<32, -1, NV, 0>(d);
            dft<32, -1, NV, 32>(d);
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int k2 = 1; k2 < 32; ++k2)
                    d[32 * h + k2] = cmul(d[32 * h + k2], SH.t2[k2 * 16 + RR(h).n3u]);
            exchange([&](int h, int j) { return RR(h).e2w(j); }, [](int, int j) { return j >> 4; },
                     [&](int h, int j) { return RR(h).e2r(j); }, [](int, int j) { return j >> 4; }, HB2);
            dft<16, -1, NV, 0>(d);
            dft<16, -1, NV, 16>(d);
            dft<16, -1, NV, 32>(d);
            dft<16, -1, NV, 48>(d);
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int j = 0; j < 16; ++j) {          // Z[k_low + 1024 k3]: blocks v and its partner
                    st2(d[32 * h + j], RR(h).vt + 1024 * j);
                    st2(d[32 * h + 16 + j], RR(h).kB + 1024 * j);
                }
        } else {
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    d[32 * h + j] = buf_ld2(rin, RR(h).vt * 8, j * 8192);
                    d[32 * h + 16 + j] = buf_ld2(rin, RR(h).kB * 8, j * 8192);
                }
            dft<16, +1, NV, 0>(d);
            dft<16, +1, NV, 16>(d);
            dft<16, +1, NV, 32>(d);
            dft<16, +1, NV, 48>(d);
            exchange([&](int h, int j) { return RR(h).e2r(j); }, [](int, int j) { return j >> 4; },
                     [&](int h, int j) { return RR(h).e2w(j); }, [](int, int j) { return j >> 4; }, HB2);
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int k2 = 1; k2 < 32; ++k2)
                    d[32 * h + k2] = cmulc(d[32 * h + k2], SH.t2[k2 * 16 + RR(h).n3u]);
            dft<32, +1, NV, 0>(d);
            dft<32, +1, NV, 32>(d);
            const T1Anch g0 = t1_load(t1q, tl);
            const T1Anch g1 = t1_load(t1q, tl + FT);
            exchange([&](int h, int j) { return RR(h).e1r(j); }, [](int h, int) { return h; },
                     [&](int h, int j) { return RR(h).e1w(j); }, [](int, int j) { return j >> 4; }, HB1);
            t1_apply<true, 0>(d, g0);
            t1_apply<true, 32>(d, g1);
            dft<32, +1, NV, 0>(d);
            dft<32, +1, NV, 32>(d);
#pragma unroll
            for (int h = 0; h < VT; ++h)
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) st2(d[32 * h + n1], 512 * n1 + tl + FT * h);
        }
    }
}

}  // namespace

// =============================================================== host side
bool ofx_fused_supported(int n_samples) {
    return n_samples == FN || ofx_fused25_supported(n_samples) ||   // 25000: ofx_fused25.hip
           ofx_fused12_supported(n_samples) ||                      // 12500: ofx_fused12.hip
           ofx_fused20_supported(n_samples) ||                      // 20000: ofx_fused20.hip
           ofx_wave_supported(n_samples) ||                         // 4096: ofx_wave.hip
           ofx_wave2_supported(n_samples);                          // 8192: ofx_wave2.hip
}

int ofx_fused_release(ofx_plan* p) {
    if (p->d_tw1) (void)hipFree(p->d_tw1);
    if (p->d_tw2) (void)hipFree(p->d_tw2);
    p->d_tw1 = p->d_tw2 = nullptr;
    if (p->d_fused_slots) (void)hipFree(p->d_fused_slots);
    if (p->d_fused_spec) (void)hipFree(p->d_fused_spec);
    if (p->d_fused_xwide) (void)hipFree(p->d_fused_xwide);
    p->d_fused_slots = p->d_fused_spec = p->d_fused_xwide = nullptr;
    p->fused_spec_bytes = 0;
    return OFX_OK;
}

static int fused_tables(ofx_plan* p) {
    if (p->d_tw1) return OFX_OK;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<float2> t1(10 * 512), t2(32 * 16 + FV);
    const int anchor_mult[10] = {1, 2, 3, 4, 5, 6, 7, 8, 16, 24};    // B1..B7, A1..A3
    for (int i = 0; i < 10; ++i)
        for (int n = 0; n < 512; ++n) {
            const long long e = ((long long)anchor_mult[i] * n) % FM;
            const double a = -PI2 * (double)e / FM;
            // float4 rows: t1a[i/2][n] = (anchor i even, anchor i odd)
            t1[((i >> 1) * 512 + n) * 2 + (i & 1)] =
                make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k2 = 0; k2 < 32; ++k2)
        for (int n3 = 0; n3 < 16; ++n3) {
            const int e = (k2 * n3) % 512;
            const double a = -PI2 * (double)e / 512.0;
            t2[k2 * 16 + n3] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int v = 0; v < FV; ++v) {      // tbase[v] = i exp(-2 pi i v / N), stored after t2
        const double a = -PI2 * (double)v / FN;
        t2[32 * 16 + v] = make_float2((float)-std::sin(a), (float)std::cos(a));
    }
    return fused_upload_tables(p, t1, t2);
}

// Build the middle-step tables of one slot from the fp64 one-sided filter.
//   d_pq layout (float4 units): [0 .. 16*512)   midW  (W_k / 2, conj(W_p) / 2)   [slot j][v]
//                               [16*512 .. +8*512) midG (g_k', g_p') as float2   [slot j][v]
//                               last entry       (W_{M/2}.x, W_{M/2}.y, g_{M/2}, 0)
// Slot j of virtual thread v pairs bin k = v + 1024 j with p = M - k (v = 0: k = 1024 j for
// j < 8 and 512 + 1024 (j - 8) above; k = 0 pairs DC with Nyquist).
int ofx_fused_prepare_slot(ofx_plan* p, int slot, const double* wf) {
    if (ofx_wave_supported(p->N)) return ofx_wave_prepare_slot(p, slot, wf);
    if (ofx_wave2_supported(p->N)) return ofx_wave2_prepare_slot(p, slot, wf);
    if (ofx_fused12_supported(p->N)) return ofx_fused12_prepare_slot(p, slot, wf);
    if (ofx_fused20_supported(p->N)) return ofx_fused20_prepare_slot(p, slot, wf);
    if (p->N != FN) return ofx_fused25_prepare_slot(p, slot, wf);
    int rc = fused_tables(p);
    if (rc) return rc;
    return fused_build_slot_tables(p, slot, wf, FM, FV, FV, 16, [](int v, int j) {
        return v != 0 ? v + 1024 * j : (j < 8 ? 1024 * j : 512 + 1024 * (j - 8));
    });
}

template <int FEAT, bool MULTI>
static int launch(ofx_plan* p, const OfxPlanDev& pd, const OfxSlotDev& sd, const FusedTabs& tabs,
                  const float* d_traces, const uint8_t* d_valid, long long n, float* d_out,
                  hipStream_t st, const FusedSlotArg* d_slots, int nslots, int nstash) {
    // (100 KiB covers the diagnostic one-workgroup-per-CU launch below as well)
    OFX_LDS_ATTR_ONCE((k_fused<FEAT, MULTI>),
                      FUSED_LDS_BYTES > 100 * 1024 ? FUSED_LDS_BYTES : 100 * 1024);
    long long grid = (long long)p->cu_count * WG_PER_CU;
    size_t lds_bytes = FUSED_LDS_BYTES;
    // diagnostic: OFX_DIAG_WGPC=1 runs one workgroup per CU (uncontended phase times)
    static const int diag_wgpc = getenv("OFX_DIAG_WGPC") ? atoi(getenv("OFX_DIAG_WGPC")) : 0;
    if (diag_wgpc == 1 && !PP) {
        grid = p->cu_count;
        lds_bytes = 100 * 1024;
    }
    if (MULTI) {
        // per-workgroup spectrum scratch (sized for the full grid, allocated once)
        const size_t need = (size_t)p->cu_count * WG_PER_CU * NHALF * NV * FT * sizeof(float2);
        if (int rcs = fused_ensure_spec(p, need)) return rcs;
    }
    if ((FEAT & 8) && !p->d_fused_xwide)   // per-workgroup stash of the bins 512 .. NS_MAX-1
        OFX_HIP(hipMalloc(&p->d_fused_xwide, (size_t)p->cu_count * WG_PER_CU * NHALF *
                                                 (NS_MAX - NLOW_MAX) * sizeof(float2)));
    if (grid * NHALF > n) grid = (n + NHALF - 1) / NHALF;
#ifdef OFX_STAMPS
    const size_t stamp_bytes =
        (size_t)grid * NHALF * OFX_STAMP_TRACES * NWAVE * 16 * sizeof(unsigned long long);
    const size_t stash_bytes = (size_t)grid * NHALF * (NS_MAX - NLOW_MAX) * sizeof(float2);
    if (p->d_fused_xwide) (void)hipFree(p->d_fused_xwide);
    p->d_fused_xwide = nullptr;
    OFX_HIP(hipMalloc(&p->d_fused_xwide, stash_bytes + stamp_bytes));
    OFX_HIP(hipMemset(p->d_fused_xwide, 0, stash_bytes + stamp_bytes));
#endif
    size_t tix = 0;
    int rc = ofx_time_begin(p, st, &tix);
    if (rc) return rc;
    hipLaunchKernelGGL((k_fused<FEAT, MULTI>), dim3((unsigned)grid), dim3(BLOCK_THREADS), lds_bytes, st, pd,
                       sd, tabs, d_traces, d_valid, n, d_out, d_slots, nslots,
                       reinterpret_cast<float2*>(p->d_fused_spec),
                       reinterpret_cast<float2*>(p->d_fused_xwide), nstash);
    rc = ofx_time_end(p, st, tix);
    if (rc) return rc;
    OFX_HIP(hipGetLastError());
#ifdef OFX_STAMPS
    if (int rcd = fused_dump_stamps(st, reinterpret_cast<char*>(p->d_fused_xwide) + stash_bytes, stamp_bytes))
        return rcd;
#endif
    return OFX_OK;
}

template <bool MULTI>
static int launch_feat(int feat, ofx_plan* p, const OfxPlanDev& pd, const OfxSlotDev& sd,
                       const FusedTabs& tabs, const float* d_traces, const uint8_t* d_valid,
                       long long n, float* d_out, hipStream_t st, const FusedSlotArg* d_slots,
                       int nslots, int nstash) {
#ifdef OFX_QUICK      // development builds: only one variant is compiled (-DOFX_QUICK=<feat>, 0 = headline;
                      // -DOFX_QUICK_MULTI: its several-slots form)
#ifdef OFX_QUICK_MULTI
    constexpr bool QM = true;
#else
    constexpr bool QM = false;
#endif
    if (feat == (OFX_QUICK + 0) && MULTI == QM)
        return launch<OFX_QUICK + 0, QM>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,
                                         nslots, nstash);
    ofx_set_error("OFX_QUICK build: only the FEAT = %d single-slot kernel exists", OFX_QUICK + 0);
    return OFX_ERR_UNSUPPORTED;
#else
#define OFX_CASE(F)                                                                            \
    case F:                                                                                    \
        return launch<F, MULTI>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,     \
                                nslots, nstash);
    switch (feat & 15) {
        OFX_CASE(0) OFX_CASE(1) OFX_CASE(2) OFX_CASE(3) OFX_CASE(4) OFX_CASE(5) OFX_CASE(6)
        OFX_CASE(7) OFX_CASE(8) OFX_CASE(9) OFX_CASE(10) OFX_CASE(11) OFX_CASE(12) OFX_CASE(13)
        OFX_CASE(14)
        default: return launch<15, MULTI>(p, pd, sd, tabs, d_traces, d_valid, n, d_out, st,
                                          d_slots, nslots, nstash);
    }
#undef OFX_CASE
#endif
}

// One launch per call: a plan with several filter slots runs them all on the shared
// forward transform (MULTI kernel); time-domain windows and bands ride along once.
int ofx_fused_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n,
                      float* d_out, hipStream_t st) {
    if (ofx_wave_supported(p->N)) return ofx_wave_process(p, d_traces, d_valid, n, d_out, st);
    if (ofx_wave2_supported(p->N)) return ofx_wave2_process(p, d_traces, d_valid, n, d_out, st);
    if (ofx_fused12_supported(p->N)) return ofx_fused12_process(p, d_traces, d_valid, n, d_out, st);
    if (ofx_fused20_supported(p->N)) return ofx_fused20_process(p, d_traces, d_valid, n, d_out, st);
    if (p->N != FN) return ofx_fused25_process(p, d_traces, d_valid, n, d_out, st);
    int rc = fused_tables(p);
    if (rc) return rc;
    FusedTabs common;
    memset(&common, 0, sizeof(common));
    common.t1 = p->d_tw1;
    common.t2 = p->d_tw2;
    common.tbase = p->d_tw2 + 32 * 16;
    {
        const double a = -6.283185307179586476925286766559 * 512.0 / FN;
        common.tb0hi = make_float2((float)-std::cos(a), (float)-std::sin(a));   // -t_512
    }
    common.midW = reinterpret_cast<const float4*>(p->d_tw1);   // never read without searches
    common.midG = p->d_tw1;

    struct G {
        enum { N = FN, ROWS = 1024, NROWS = 32, LDS_BINS = NLOW_MAX,
               MAX_BINS = (VT == 2) ? NS_MAX : NLOW_MAX, MIDG_OFF = 16 * FV };
        using Tabs = FusedTabs;
        using SlotArg = FusedSlotArg;
    };
    return fused_process_plan<G>(p, common, st, [&](bool multi, int feat, const OfxPlanDev& pd,
                                                    const OfxSlotDev& sd, const FusedTabs& tabs,
                                                    const FusedSlotArg* d_slots, int nslots, int nstash) {
        return multi ? launch_feat<true>(feat, p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,
                                         nslots, nstash)
                     : launch_feat<false>(feat, p, pd, sd, tabs, d_traces, d_valid, n, d_out, st, d_slots,
                                          nslots, nstash);
    });
}

// ---- the transform on its own (rows of 16384 complex points; used by the N x M engine)
int ofx_fused_fft_create(int n_complex, int device, void** out) {
    if (n_complex != FM || VT != 2 || PP) return OFX_ERR_UNSUPPORTED;
    return fused_fft_create(device, out, fused_tables);
}
void ofx_fused_fft_destroy(void* h) { fused_fft_destroy(h); }
int ofx_fused_fft_exec(void* h, bool forward, const float2* in, float2* out, long long rows,
                       hipStream_t st) {
    return fused_fft_exec<&k_fft32<true>, &k_fft32<false>>(h, forward, in, out, rows, st, WG_PER_CU, FT,
                                                           sizeof(FusedShared));
}
