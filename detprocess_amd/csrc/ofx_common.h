// ofx_common.h -- plan structures shared by the C ABI and the two engines.
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ofx.h"

#define OFX_SENTINEL (-999999.0f)

void ofx_set_error(const char* fmt, ...);

#define OFX_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            ofx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,       \
                          hipGetErrorString(e_));                              \
            return OFX_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

#define OFX_FFT(expr)                                                          \
    do {                                                                       \
        rocfft_status s_ = (expr);                                             \
        if (s_ != rocfft_status_success) {                                     \
            ofx_set_error("%s:%d: %s -> rocfft status %d", __FILE__, __LINE__, \
                          #expr, (int)s_);                                     \
            return OFX_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

// Process-wide one-time initialisation, safe when plans are created from several threads.
int ofx_rocfft_setup_once();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel instantiation
#define OFX_LDS_ATTR_ONCE(kernel, bytes)                                                   \
    do {                                                                                   \
        static std::once_flag once_;                                                       \
        static hipError_t err_ = hipSuccess;                                               \
        std::call_once(once_, [&] {                                                        \
            err_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                       (int)(bytes));                                      \
        });                                                                                \
        OFX_HIP(err_);                                                                     \
    } while (0)

// ---------------------------------------------------------------- device view
struct OfxSearchDev {
    int kind;      // OFX_SEARCH_*
    int lo, hi;    // half-open rolled range
    int outside;   // search complement of [lo,hi)
    int nlow;      // one-sided bins k = 0..nlow-1 have |f_k| <= lowchi2_fcutoff
    int out_off;   // float offset of this search's record in the output row
    int interp;    // OFX_SEARCH_DELAY_INTERP: refine around the discrete minimum
};

struct OfxSlotDev {
    const float2* wf;   // [K]   filter: A = C2R(wf * V)
    const float* g;     // [K]   chi2 weights
    const float2* s;    // [K]   template FFT (for lowchi2)
    const float4* pq;   // [M]   fused engine: (P_k, Q_k) of the packed real-FFT filter
    float norm;
    float tres_sum;
    float ampres;
    int n_search;
    OfxSearchDev search[OFX_MAX_SEARCHES];
};

struct OfxTdWinDev {
    int lo, hi;     // end-exclusive slice
    int out_off;
    // register rows of the fused kernels (set per launch by the engine, whose row length they
    // depend on): bit n1 of `full` = row n1 lies inside the slice, of `edge` = it is cut by it
    unsigned full = 0, edge = 0;
};

struct OfxBandDev {
    int k_lo, k_hi;     // one-sided bins [k_lo, k_hi)
    int out_off;
};

struct OfxPlanDev {
    int N, K, pre;
    float fs, inv_fs;
    int row;                 // floats per output row
    int n_channels, n_terms;
    int chan[OFX_MAX_TERMS];
    float weight[OFX_MAX_TERMS];
    int n_tdwin;
    OfxTdWinDev tdw[OFX_MAX_TDWIN];
    int n_bands;
    OfxBandDev band[OFX_MAX_BANDS];
};

// ------------------------------------------------------------------ host plan
struct OfxSlotHost {
    bool set = false;
    float2* d_wf = nullptr;
    float* d_g = nullptr;
    float2* d_s = nullptr;
    float4* d_pq = nullptr;
    double norm = 0, tres_sum = 0;
    float wq_x = 0, wq_y = 0, gq = 0;   // fused engine: self-paired bin k = M/2
    std::vector<double> g_host;     // kept to count low-frequency bins
    std::vector<double> wf_host;    // [2K] interleaved, kept for the LDS engine's pair tables
    std::vector<OfxSearchDev> searches;
};

struct OfxFftPlans {
    rocfft_plan r2c = nullptr, c2r = nullptr;
    rocfft_execution_info info_r2c = nullptr, info_c2r = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
};

// ofx_lds.hip: the LDS-resident mixed-radix transform on its own (batched, natural-order rows;
// lengths 2^a 3^b 5^c that fit in LDS).  create returns OFX_ERR_UNSUPPORTED for other lengths.
struct OfxLdsFft;
// allow_reg: rows of 12500 / 6250 points may go to the register-resident transform of
// ofx_fused25.hip (the N x M engine); the ROCFFT engine, which the tests use as the independent
// cross-check of the fused kernels, keeps the LDS transform
int ofx_ldsfft_create(int n_complex, int device, OfxLdsFft** out, bool allow_reg = false);
void ofx_ldsfft_destroy(OfxLdsFft* f);
int ofx_ldsfft_exec(OfxLdsFft* f, bool forward, const float2* in, float2* out, long long rows,
                    hipStream_t st);

struct ofx_plan {
    int N = 0, K = 0, pre = 0;
    double fs = 0;
    int max_batch = 0;
    int device = 0;
    int engine = OFX_ENGINE_ROCFFT;
    bool engine_auto = false;            // created with OFX_ENGINE_AUTO: may fall back per call
    int n_channels = 1, n_terms = 1;
    int chan[OFX_MAX_TERMS] = {0};
    double weight[OFX_MAX_TERMS] = {1.0};
    OfxSlotHost slot[OFX_MAX_SLOTS];
    std::vector<OfxTdWinDev> tdwin;
    std::vector<OfxBandDev> bands;
    int cu_count = 256;

    // ROCFFT engine buffers (lazy)
    std::map<int, OfxFftPlans> fft;      // keyed by batch
    OfxLdsFft* ldsfft = nullptr;  // non-power-of-two lengths the LDS transform handles
    bool ldsfft_tried = false;
    float* d_trace = nullptr;            // [max_batch, N] combined trace (if needed)
    float2* d_spec = nullptr;            // [max_batch, K]
    float2* d_filt = nullptr;            // [max_batch, K]
    float* d_amp = nullptr;              // [max_batch, N]
    float* d_chi0 = nullptr;             // [max_batch]
    float2* d_vlow = nullptr;            // [max_batch, vlow_cap] low bins of V (lowchi2, psd_amp)
    int vlow_cap = 0;
    // staging for host buffers
    float* d_stage_in = nullptr;
    size_t stage_in_floats = 0;
    uint8_t* d_stage_valid = nullptr;
    size_t stage_valid_elems = 0;
    float* d_stage_out = nullptr;
    size_t stage_out_floats = 0;

    // fused engine tables
    float2* d_tw1 = nullptr;             // stage-1 inter-stage twiddles
    float2* d_tw2 = nullptr;
    int16_t* d_adc = nullptr;            // ofx_process_adc: staged stream, trigger indices
    size_t adc_elems = 0;
    long long* d_trig = nullptr;
    size_t trig_elems = 0;
    void* d_lds_tw = nullptr;            // LDS engine: per-stage twiddle tables, slot table
    int lds_tw_level = 0;                // ... and the table level they were built for
    void* d_lds_slots = nullptr;
    void* d_lds_pos = nullptr;           // ... per-slot pair tables (LdsPair)
    void* d_lds_spec = nullptr;          // ... per-workgroup spectrum scratch (several slots)
    size_t lds_spec_bytes = 0;
    unsigned long long filter_stamp = 0; // bumped by set_filter / add_search / reset
    unsigned long long lds_pair_stamp = ~0ull;
    int lds_pair_slots = 0;
    unsigned long long lds_slot_stamp = ~0ull, fused_slot_stamp = ~0ull;
    std::vector<unsigned char> h_slot_args;   // host copy of the slot table of the last launch
                                              // (kept alive: the upload is asynchronous)
    void* d_fused_slots = nullptr;       // FUSED multi-slot launches: slot table ...
    void* d_fused_spec = nullptr;        // ... and per-workgroup spectrum scratch
    size_t fused_spec_bytes = 0;
    void* d_fused_xwide = nullptr;       // ... and per-workgroup stash of the bins 512 .. 4095 of 2 X_k

    // timing of the dominant kernel
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t ev_used = 0;
    double t_acc_ms = 0;
    long long t_launches = 0;
};

int ofx_row_floats(const ofx_plan* p);
void ofx_fill_plan_dev(const ofx_plan* p, OfxPlanDev* d);
void ofx_fill_slot_dev(const ofx_plan* p, int slot, OfxSlotDev* d);

// engines
int ofx_rocfft_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                       long long n, float* d_out, hipStream_t st);
int ofx_rocfft_release(ofx_plan* p);
bool ofx_fused_supported(int n_samples);
int ofx_fused_prepare_slot(ofx_plan* p, int slot, const double* wf);
int ofx_fused_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                      long long n, float* d_out, hipStream_t st);
int ofx_fused_release(ofx_plan* p);
// ofx_fused25.hip: the register-resident kernel for 25000-sample traces (dispatched by ofx_fused_*)
bool ofx_fused25_supported(int n_samples);
int ofx_fused25_prepare_slot(ofx_plan* p, int slot, const double* wf);
int ofx_fused25_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                        long long n, float* d_out, hipStream_t st);
int ofx_fused_fft_create(int n_complex, int device, void** h);        // rows of 16384 complex points
void ofx_fused_fft_destroy(void* h);
int ofx_fused_fft_exec(void* h, bool forward, const float2* in, float2* out, long long rows,
                       hipStream_t st);
int ofx_fused25_fft_create(int n_complex, int device, void** h);      // rows of 12500 complex points
void ofx_fused25_fft_destroy(void* h);
int ofx_fused25_fft_exec(void* h, bool forward, const float2* in, float2* out, long long rows,
                         hipStream_t st);
int ofx_fused12_fft_create(int n_complex, int device, void** h);      // rows of 6250 complex points
void ofx_fused12_fft_destroy(void* h);
int ofx_fused12_fft_exec(void* h, bool forward, const float2* in, float2* out, long long rows,
                         hipStream_t st);
// ofx_fused20.hip: the same source built for 20000-sample traces
bool ofx_fused20_supported(int n_samples);
int ofx_fused20_prepare_slot(ofx_plan* p, int slot, const double* wf);
int ofx_fused20_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                        long long n, float* d_out, hipStream_t st);
int ofx_fused20_fft_create(int n_complex, int device, void** h);      // rows of 10000 complex points
void ofx_fused20_fft_destroy(void* h);
int ofx_fused20_fft_exec(void* h, bool forward, const float2* in, float2* out, long long rows,
                         hipStream_t st);
// ofx_fused12.hip: the same source built for 12500-sample traces
bool ofx_fused12_supported(int n_samples);
int ofx_fused12_prepare_slot(ofx_plan* p, int slot, const double* wf);
int ofx_fused12_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                        long long n, float* d_out, hipStream_t st);
// ofx_wave.hip: one wave per trace, 4096-sample traces
bool ofx_wave_supported(int n_samples);
int ofx_wave_prepare_slot(ofx_plan* p, int slot, const double* wf);
int ofx_wave_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n,
                     float* d_out, hipStream_t st);
// ofx_wave2.hip: two waves per trace, 8192-sample traces
bool ofx_wave2_supported(int n_samples);
int ofx_wave2_prepare_slot(ofx_plan* p, int slot, const double* wf);
int ofx_wave2_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid, long long n,
                      float* d_out, hipStream_t st);
bool ofx_lds_supported(int n_samples);
int ofx_lds_process(ofx_plan* p, const float* d_traces, const uint8_t* d_valid,
                    long long n, float* d_out, hipStream_t st);
int ofx_lds_release(ofx_plan* p);

// timing helpers
int ofx_time_begin(ofx_plan* p, hipStream_t st, size_t* idx);
int ofx_time_end(ofx_plan* p, hipStream_t st, size_t idx);

// ofx_ingest.hip: windows of int16 streams -> float32 [nb, C, N] events + valid mask
int ofx_cut_launch(const int16_t* d_adc, long long n_stream, int n_channels, int n_samples,
                   int n_pretrigger, const long long* d_trig, long long nb, const float* scale,
                   const float* offset, float* d_events, uint8_t* d_valid, hipStream_t st);
