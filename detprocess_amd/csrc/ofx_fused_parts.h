// ofx_fused_parts.h -- what the register-resident kernels k_fused (ofx_fused.hip, 32768 samples) and
// k_fused25 (ofx_fused25.hip: 25000 / 20000 / 12500 samples) share at file scope: buffer-descriptor
// loads, the pairwise middle step, small complex helpers.  Included inside each file's anonymous
// namespace.  The blocks the two kernels share INSIDE their bodies are the fragments
// ofx_fused_*.inc (time-domain window end points and finalisation, resolve, the row write): textual
// inclusion, because a workgroup's trace lives in 128 VGPRs per thread and every one of these blocks is
// scheduled and register-allocated together with its surroundings -- the instruction streams of all
// four objects are unchanged by the split (checked when it was made, round 3).
#pragma once

// Buffer loads: descriptor in SGPRs, one 32-bit VGPR byte offset per lane, the row offset in an
// SGPR -- no per-row 64-bit VGPR addresses to keep alive.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
#ifdef ABL_NOTAB
    return make_float4(0.5f + voff * 1e-9f, 0.25f, 0.125f + soff * 1e-9f, 0.7f);
#endif
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                       __uint_as_float(v.w));
}
__device__ __forceinline__ cpx buf_ld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return mk(__uint_as_float(v.x), __uint_as_float(v.y));
}
// The thread index from scratch-free sources: 64 x wave (a scalar) + the lane count of v_mbcnt.  The phases of the
// kernels re-derive their addresses from a fresh copy of the thread index (so that LICM does not hoist and spill
// them); taking that copy from the live `tid` kept `tid` itself in a spill slot, reloaded -- behind an s_waitcnt
// vmcnt(0) -- six times per trace, once in front of the 64 loads of the next trace.
__device__ __forceinline__ int ofx_fresh_tid(int wave_base) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return wave_base + l;
}
__device__ __forceinline__ cpx lo2(const float4& q) { return mk(q.x, q.y); }
__device__ __forceinline__ cpx hi2(const float4& q) { return mk(q.z, q.w); }
__device__ __forceinline__ cpx cconj(cpx z) { return z * mk(1.0f, -1.0f); }

// The pairwise middle step on one (Z_k, Z_p) slot, p = M - k:
// in zk = Z_k, zp = Z_p; out zk = Z'_k, zp = Z'_p; xk2 = 2 X_k, xp2 = 2 conj(X_p).
// T = i t_k, tw = (W_k / 2, conj(W_p) / 2), g = (g_k', g_p').
__device__ __forceinline__ void mid_slot(cpx& zk, cpx& zp, const cpx T, const float4 tw,
                                         const cpx g, cpx& xk2, cpx& xp2, cpx& chi) {
    const cpx wk = lo2(tw), wp = hi2(tw);
    const cpx u = pfma(zp, mk(1.0f, -1.0f), zk);             // Z_k + conj(Z_p)
    const cpx w = pfma(zp, mk(-1.0f, 1.0f), zk);             // Z_k - conj(Z_p)
    const cpx sv = cmul(w, T);
    xk2 = u - sv;
    xp2 = u + sv;
    chi = pfma(xk2 * xk2, g.xx, chi);
    chi = pfma(xp2 * xp2, g.yy, chi);
    const cpx yk = cmul(xk2, wk);
    const cpx yp = cmul(xp2, wp);
    const cpx sg = yk + yp;
    const cpx df = yk - yp;
    const cpx q = cmulc(df, T);
    zk = sg - q;
    zp = conj_sum(sg, q);
}
