// ofx_fused_stamps.h -- phase markers of the register-resident kernels (ofx_fused.hip, ofx_fused25.hip).
// Product builds: STAMP(i) is an assembly comment (";ofxphase i") that marks the phases in the ISA
// (tools/isa_phases.py, tools/isa_hazards.py reports carry them); SUBSTAMP / TSTAMP are empty.
// Included inside each file's anonymous namespace, after NWAVE is defined; the diagnostic build expects
// `stamp_it` and `stamp_base` in the kernel.
#pragma once

// Phase markers: an assembly comment (";ofxphase i") to find the phases in the ISA
// (hipcc -S; tools/isa_phases.py counts instructions per phase).
#ifndef OFX_STAMPS
#define STAMP(i) asm volatile(";ofxphase " #i)
#define SUBSTAMP(i)
#else
#define SUBSTAMP(i) STAMP(i)
// Diagnostic build (-DOFX_STAMPS, tools/phase_timeline.py): every wave writes the shader clock at
// every phase marker of its first OFX_STAMP_TRACES traces into the buffer passed in place of
// `xwide` ([workgroup][trace][wave][16]; slot 13: HW_ID, slot 14: XCC_ID) with SCALAR stores --
// no branch, no exec-mask change, so the scheduling regions of the product build stay as they
// are -- so that the phases of the two workgroups of a CU can be laid over each other.  No
// output depends on the stamps.
#define OFX_STAMP_TRACES 40
__device__ __forceinline__ void ofx_stamp(unsigned long long* p) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0"
                 : "=&s"(t) : "s"(p));
}
// slot 15: the constant 100 MHz counter at stamp 0 -- shader clock in the kernel =
// d(s_memtime) / d(s_memrealtime) x 100 MHz between the stamps 0 of consecutive traces
// (MI355X_MICROARCH.md, DVFS note (6); tools/phase_timeline.py reports it as clock_mhz)
__device__ __forceinline__ void ofx_stamp_rt(unsigned long long* p) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x78"
                 : "=&s"(t) : "s"(p));
}
__device__ __forceinline__ void ofx_stamp_id(unsigned long long* p) {
    unsigned a, b;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\n\t"
                 "s_store_dword %0, %2, 0x68\n\ts_store_dword %1, %2, 0x70"
                 : "=&s"(a), "=&s"(b) : "s"(p));
}
// -DOFX_TAILSTAMPS (with -DOFX_STAMPS): the stamps 2..9 of the transform phases are dropped and their
// slots carry the sub-phases of the tail instead (TSTAMP(2..9); tools/dev_tail_timeline.py).
#ifdef OFX_TAILSTAMPS
constexpr bool TAILMODE = true;
#else
constexpr bool TAILMODE = false;
#endif
#define STAMP_AT(i)                                                                           \
    do {                                                                                      \
        asm volatile(";ofxphase " #i);                                                        \
        {   /* no branch: traces beyond the last slot keep overwriting it */                  \
            const int si_ = stamp_it < OFX_STAMP_TRACES - 1 ? stamp_it : OFX_STAMP_TRACES - 1; \
            unsigned long long* sb_ = stamp_base + (size_t)si_ * (NWAVE * 16);                \
            ofx_stamp(sb_ + (i));                                                             \
            if ((i) == 0) ofx_stamp_id(sb_);                                                  \
            if ((i) == 0) ofx_stamp_rt(sb_);                                                  \
            if ((i) == 12) ++stamp_it;                                                        \
        }                                                                                     \
    } while (0)
#define STAMP(i)                                                                              \
    do {                                                                                      \
        if constexpr (!(TAILMODE && (i) >= 2 && (i) <= 9)) STAMP_AT(i);                       \
    } while (0)
#define TSTAMP(i)                                                                             \
    do {                                                                                      \
        if constexpr (TAILMODE) STAMP_AT(i);                                                  \
    } while (0)
#endif
#ifndef TSTAMP
#define TSTAMP(i)
#endif
