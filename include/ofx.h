/*
 * ofx.h -- C ABI of libofx.so, the MI355X (gfx950) optimal-filter feature engine.
 *
 * This is the drop-in boundary for the per-event hot path of
 * spice-herald/detprocess.  Nothing in the reference calls C (it is 100 %
 * Python); each entry point below names the reference interface whose WORK it
 * replaces.  The Python shim (detprocess_amd/_lib.py) binds exactly these
 * symbols with ctypes; INTEGRATION.md shows the stub a detprocess maintainer
 * would add.
 *
 * Conventions: status-int returns (0 = OFX_OK), no exceptions cross the ABI,
 * caller-owned buffers, plain pointers and sizes, a plan is not thread-safe
 * (one plan per stream).  Device pointers are HIP device pointers on the plan's
 * device; `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef OFX_H
#define OFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ofx_plan ofx_plan;

enum {
    OFX_OK = 0,
    OFX_ERR_ARG = 1,      /* bad argument (the reference raises ValueError) */
    OFX_ERR_HIP = 2,      /* HIP runtime / rocFFT failure                   */
    OFX_ERR_STATE = 3,    /* plan not fully configured                      */
    OFX_ERR_UNSUPPORTED = 4
};

enum { OFX_MEM_HOST = 0, OFX_MEM_DEVICE = 1 };

/* engines: FUSED = one persistent LDS-resident FFT kernel per trace
 *          ROCFFT = rocFFT R2C -> filter kernel -> rocFFT C2R -> arg-max kernel
 *          AUTO = FUSED where the trace length is supported, else ROCFFT     */
enum { OFX_ENGINE_AUTO = 0,     /* FUSED if n_samples is 32768, 25000, 20000, 16384, 12500, 8192 or 4096, else LDS if it applies, else ROCFFT */
       OFX_ENGINE_FUSED = 1,    /* register/LDS-resident kernels: n_samples == 32768 (k_fused),
                                   25000 and 12500 (k_fused25 / k_fused12: the 20 ms and 10 ms traces
                                   of the reference's examples at 1.25 MHz), 20000 (k_fused20) and
                                   4096 / 8192 / 16384 (k_wave / k_wave2, one / two / four waves per trace); lowchi2
                                   cut-offs and psd_amp bands up to 62 kHz = 1250 / 1000 / 625 bins (4096 /
                                   8192 / 16384 samples: 256 / 512 / 1024 bins = 78 kHz), beyond that AUTO plans fall back */
       OFX_ENGINE_ROCFFT = 2,   /* rocFFT pipeline, any even n_samples                         */
       OFX_ENGINE_LDS = 3 };    /* LDS-resident kernel, n_samples/2 = 2^a 3^b 5^c, <= 34816    */

/* search kinds (one per of1x1 algorithm instance) */
enum {
    OFX_SEARCH_NODELAY = 0,   /* algorithms.py:277-350  of1x1_nodelay        */
    OFX_SEARCH_DELAY = 1,     /* algorithms.py:354-432 / 435-570             */
    OFX_SEARCH_DELAY_INTERP = 2  /* same with interpolate=True (algorithms.py:357, 443):
                                    3-point parabolic refinement of t0 / amp / chi2 around
                                    the discrete minimum; lowchi2 at the refined (amp, t0) */
};

#define OFX_MAX_SLOTS 8       /* (template_tag, csd_tag) filters per plan     */
#define OFX_MAX_SEARCHES 8    /* of1x1 algorithm instances per filter slot    */
#define OFX_MAX_TDWIN 8       /* baseline/integral/min/max windows per plan   */
#define OFX_MAX_BANDS 16      /* psd_amp frequency bands per plan             */
#define OFX_MAX_TERMS 8       /* channels combined by '+' / '-' on load       */

/* per-search output record, floats, in this order */
#define OFX_SEARCH_FLOATS 8
enum {
    OFX_COL_AMP = 0,          /* amp_<name>                                   */
    OFX_COL_T0 = 1,           /* t0_<name>  [s]  ((index - pretrigger)/fs)    */
    OFX_COL_CHI2 = 2,         /* chi2_<name>                                  */
    OFX_COL_LOWCHI2 = 3,      /* lowchi2_<name>                               */
    OFX_COL_CHI2NOPULSE = 4,  /* chi2nopulse_<name>                           */
    OFX_COL_AMPRES = 5,       /* ampres_<name>  (constant of the filter)      */
    OFX_COL_TIMERES = 6,      /* timeres_<name>                               */
    OFX_COL_INDEX = 7         /* rolled bin index of the fit, as a float      */
};
/* per-time-domain-window output record, floats, in this order */
#define OFX_TDWIN_FLOATS 8
enum {
    OFX_TD_BASELINE = 0,      /* algorithms.py:698  mean(trace[lo:hi])        */
    OFX_TD_INTEGRAL = 1,      /* algorithms.py:759  trapz(trace[lo:hi])/fs    */
    OFX_TD_MAXIMUM = 2,       /* algorithms.py:818                            */
    OFX_TD_MINIMUM = 3,       /* algorithms.py:879                            */
    OFX_TD_SUM = 4,           /* sum(trace[lo:hi])      } inputs of           */
    OFX_TD_SUMSQ = 5,         /* sum(trace[lo:hi]**2)   } energyabsorbed,     */
    OFX_TD_FIRST = 6,         /* trace[lo]              } algorithms.py:938-943 */
    OFX_TD_LAST = 7           /* trace[hi-1]            }                     */
};
/* per-band output: ONE float = mean over bins [k_lo, k_hi) of
 * sqrt(folded PSD of the event) -- psd_amp, algorithms.py:1013-1038 */
#define OFX_BAND_FLOATS 1

/* last error text of the calling thread ("" if none). */
const char* ofx_last_error(void);

/* library / device facts: fills "version;arch;cu_count" style text. */
int ofx_device_info(int device, char* buf, size_t buflen);

/*
 * Create a plan for traces of n_samples at rate fs with n_pretrigger samples
 * before the trigger.  Replaces qp.OFBase(sample_rate) construction keyed by
 * (nb_samples, nb_pretrigger_samples, csd tag) --
 * detprocess/process/processing_data.py:274-286.
 * max_batch bounds the traces per ofx_process call chunk (work-buffer size for
 * the ROCFFT engine; ignored by FUSED).
 */
int ofx_plan_create(ofx_plan** plan, int n_samples, int n_pretrigger, double fs,
                    int max_batch, int device, int engine);
int ofx_plan_destroy(ofx_plan* plan);

/* which engine the plan resolved to (OFX_ENGINE_FUSED / OFX_ENGINE_LDS / OFX_ENGINE_ROCFFT). */
int ofx_plan_engine(const ofx_plan* plan);

/*
 * Upload one precomputed optimal filter (host, fp64, one-sided K = n/2+1 bins).
 * Replaces OFBase.set_csd + add_template + calc_phi --
 * processing_data.py:321-326, 369-381 (the one-time precompute stays on the
 * host; this call only rounds it to fp32 device tables).
 *   wf[2K]   interleaved re,im of conj(S_k)/J_k/(N fs)/norm  -> A = C2R(wf*V)
 *   g[K]     chi2 weights 1/(J_k N fs)   (0 where J = inf)
 *   s[2K]    interleaved re,im of the template FFT S_k (NumPy, unnormalised)
 *   norm, tres_sum: scalars of SURVEY.md Appendix A
 */
int ofx_plan_set_filter(ofx_plan* plan, int slot, const double* wf,
                        const double* g, const double* s, double norm,
                        double tres_sum);

/*
 * Register one of1x1 algorithm instance on a filter slot.  Replaces the
 * per-event qp.OF1x1(...).calc(...) arguments -- algorithms.py:336-338,
 * 414-418, 538-547.  [lo, hi) is the half-open range of ROLLED bin indices
 * searched (outside != 0: its complement); ignored for OFX_SEARCH_NODELAY.
 * lowchi2_fcutoff: bins with |f_k| <= cutoff enter lowchi2.  Engine limits on that count
 * (checked at ofx_process, OFX_ERR_UNSUPPORTED; an OFX_ENGINE_AUTO plan then runs the call on
 * the LDS engine where that one carries the length, else on the ROCFFT engine): FUSED 4096 bins
 * (156 kHz at 32768 samples / 1.25 MHz; 1250 / 1000 / 1024 / 625 / 512 / 256 bins at 25000 / 20000 / 16384 / 12500 / 8192 / 4096; the reference
 * example's 50 kHz = 1311 bins, examples/processing/process_example.yaml:113), LDS 1024 bins,
 * ROCFFT none.  The same limits hold for the upper bin of ofx_plan_add_band.
 * Returns the search id (>= 0) or a negative error.
 */
int ofx_plan_add_search(ofx_plan* plan, int slot, int kind, int lo, int hi,
                        int outside, double lowchi2_fcutoff);

/*
 * Register one time-domain window: baseline/integral/maximum/minimum of
 * trace[lo:hi] (end-exclusive) -- algorithms.py:698, 759, 818, 879.
 * Returns the window id (>= 0) or a negative error.
 */
int ofx_plan_add_tdwindow(ofx_plan* plan, int lo, int hi);

/*
 * Register one psd_amp band: one-sided FFT bins [k_lo, k_hi), 1 <= k_lo < k_hi <=
 * n/2+1 (bin 0 = DC is dropped by the reference, algorithms.py:1018-1021).  The
 * feature is mean_k sqrt(w_k |V_k|^2 / (N fs)), w_k = 2 (1 at Nyquist): the folded
 * PSD of algorithms.py:1013-1016.  Returns the band id (>= 0) or a negative error.
 */
int ofx_plan_add_band(ofx_plan* plan, int k_lo, int k_hi);

/*
 * Channel algebra on load: the processed trace is sum_j weight[j] *
 * event[chan_index[j]] over the n_channels rows of each event.  Replaces
 * ProcessingData.get_channel_trace -- processing_data.py:1033-1047.
 * Default: n_channels = 1, one term (row 0, weight 1).
 */
int ofx_plan_set_channels(ofx_plan* plan, int n_channels, int n_terms,
                          const int* chan_index, const double* weight);

/* drop all filters / searches / windows (keeps buffers). */
int ofx_plan_reset(ofx_plan* plan);

/* floats per output row:  sum over slots of n_search*OFX_SEARCH_FLOATS,
 * then n_tdwin*OFX_TDWIN_FLOATS, then n_bands*OFX_BAND_FLOATS.  Column offsets
 * via the calls below. */
int ofx_plan_row_floats(const ofx_plan* plan);
int ofx_plan_search_offset(const ofx_plan* plan, int slot, int search);
int ofx_plan_tdwindow_offset(const ofx_plan* plan, int window);
int ofx_plan_band_offset(const ofx_plan* plan, int band);

/*
 * Process n_traces events.  traces: float32 [n_traces, n_channels, n_samples],
 * contiguous.  valid: optional uint8 [n_traces] (same memory kind as traces);
 * rows with valid == 0 are not processed and every column is -999999.0
 * (algorithms.py:319-327, 398-407, 517-529, 683-688).  out: float32
 * [n_traces, row_floats].  Replaces, per event, update_signal_OF
 * (processing_data.py:712-772) and every FeatureExtractors.of1x1_* /
 * baseline / integral / maximum / minimum call of features.py:692-851.
 * Asynchronous on `stream` when both buffers are device memory.
 */
int ofx_process(ofx_plan* plan, const float* traces, const uint8_t* valid,
                long long n_traces, int traces_mem, float* out, int out_mem,
                void* stream);

/*
 * Cut-and-convert front end: the events of a dump are windows of continuous
 * raw-data streams.  adc: int16 [n_channels, n_stream] (channel-major, the
 * plan's n_channels), host or device memory; trigger_index: HOST int64
 * [n_events], the sample index of each trigger in the stream.  Event b is
 *   amps[c, j] = (float)adc[c, trigger_index[b] - n_pretrigger + j] * scale[c] + offset[c]
 * for j in [0, n_samples), the product and the sum each rounded to float32 (the
 * ADC-to-amps conversion pytesio's reader applies with adctoamp=True; its
 * coefficients come from the file's detector_config and are passed in here).
 * A window that does not fit in the stream gives a row of -999999.0 -- "trace
 * could not be cut", processing_data.py:640-656, 734-736.  The cut windows are
 * then processed exactly as by ofx_process.  Replaces
 * H5Reader.read_single_event(trigger_index=, trace_length_samples=,
 * pretrigger_length_samples=, adctoamp=True) + the truncation of
 * processing_data.py:640-656, 674-684: only 2 bytes per stream sample cross
 * PCIe, however many (overlapping) windows are cut from it.
 */
int ofx_process_adc(ofx_plan* plan, const int16_t* adc, long long n_stream, int adc_mem,
                    const long long* trigger_index, long long n_events,
                    const double* scale, const double* offset,
                    float* out, int out_mem, void* stream);

/*
 * Device-side synthetic event generator (bench / tests): fills
 * traces[n_traces, n_samples] with  amp_b * roll(template, delay_b) + sigma *
 * white Gaussian noise, counter-based (seed, global trace index) so any shard
 * of a run is reproducible.  amp/delay per trace are written to
 * truth[n_traces, 2] if not NULL.  template_td: device float[n_samples].
 */
int ofx_synth_traces(float* traces, float* truth, long long n_traces,
                     long long first_index, int n_samples,
                     const float* template_td, float sigma, float amp_lo,
                     float amp_hi, float pulse_fraction, int max_delay,
                     unsigned long long seed, void* stream);

/*
 * Same events with COLOURED Gaussian noise consistent with a PSD:
 * noise = irfft(sqrt(J N fs / 2) (xi1 + i xi2)).  noise_amp: device float[n/2+1] =
 * sqrt(J_k N fs / 2) / N (the 1/N of NumPy's irfft folded in).  Uses a cached rocFFT
 * C2R plan and spectrum buffer; ofx_synth_release() frees them.
 */
int ofx_synth_traces_psd(float* traces, float* truth, long long n_traces,
                         long long first_index, int n_samples, const float* template_td,
                         const float* noise_amp, float amp_lo, float amp_hi,
                         float pulse_fraction, int max_delay, unsigned long long seed,
                         void* stream);
int ofx_synth_release(void);

/* average GPU time (ms) of the dominant kernel over the launches recorded since
 * the last call, measured with HIP events on the launch stream; resets the
 * accumulator.  n_launches receives the launch count. */
int ofx_plan_kernel_time(ofx_plan* plan, double* avg_ms, long long* n_launches);
int ofx_plan_enable_timing(ofx_plan* plan, int enable);

/* ------------------------------------------------------------------------
 * Continuous-data optimal-filter trigger, one channel x one amplitude
 * (SURVEY.md section 8f rank 3).  Replaces, for that case,
 * OptimumFilterTrigger.update_trace (detprocess/core/oftrigger.py:588-679) and
 * the threshold / range-merging / arg-max part of find_triggers_once
 * (oftrigger.py:884-1035, static pile-up window).
 * --------------------------------------------------------------------- */
typedef struct ofx_trigger ofx_trigger;

/*
 * phi_td: fp64 [n_samples], the time-domain optimal filter ifft(phi).real with
 * phi[0] = 0 (oftrigger.py:486-489).  vscale: filtered = conv(trace, phi_td) /
 * vscale (the iweight matrix times the scale of phi); w: the 1x1 weight matrix,
 * delta_chi2 = filtered^2 * w (oftrigger.py:663-671).
 */
int ofx_trigger_create(ofx_trigger** out, int n_samples, int n_pretrigger, double fs,
                       const double* phi_td, double vscale, double w, int device);
int ofx_trigger_destroy(ofx_trigger* trig);

/*
 * N channels x M amplitudes (oftrigger.py:407-499, n_chan, n_amp <= 4): phi_td fp64
 * [n_chan][n_amp][n_samples] = ifft(phi).real per (channel, amplitude) with the DC bin zeroed;
 * iw [n_amp][n_amp] maps the summed convolutions onto amplitudes, filtered = iw V_td
 * (the iweight matrix times the scale of phi), w [n_amp][n_amp] is the weight matrix,
 * delta_chi2 = filtered^T w filtered (oftrigger.py:656-671).
 */
int ofx_trigger_create_nxm(ofx_trigger** out, int n_samples, int n_pretrigger, double fs,
                           int n_chan, int n_amp, const double* phi_td, const double* iw,
                           const double* w, int device);

/*
 * update_trace: FIR-filter a continuous stream ('same'-mode linear convolution,
 * scipy.signal.oaconvolve in the reference; overlap-save with batched rocFFT
 * here) and form delta chi2; padding != 0 zeroes delta chi2 within n_samples of
 * both ends exactly as oftrigger.py:674-679.  stream: n values, float32
 * (dtype 0) or int16 (dtype 1, amps = adc * scale + offset), host or device.
 * The filtered and delta-chi2 traces stay on the device inside the object;
 * ofx_trigger_get_traces copies them out (either pointer may be NULL).
 */
int ofx_trigger_update_trace(ofx_trigger* trig, const void* stream_data, int dtype,
                             long long n, int mem, double scale, double offset, int padding,
                             void* stream);
/* N x M form: stream_data is [n_chan][n]; scale / offset per channel (read for dtype 1);
 * the filtered traces are kept as [n_amp][n] */
int ofx_trigger_update_traces(ofx_trigger* trig, const void* stream_data, int dtype, long long n,
                              int mem, const double* scale, const double* offset, int padding,
                              void* stream);
int ofx_trigger_get_traces(ofx_trigger* trig, float* filtered, float* delta_chi2, int mem,
                           void* stream);

/*
 * find_triggers_once: samples with delta chi2 > chi2_threshold are grouped into
 * ranges whose consecutive members are at most pileup_window samples apart
 * (_getchangeslessthanthresh, oftrigger.py:29-77); each range yields its arg-max
 * (first maximum).  Outputs (HOST arrays of capacity max_triggers, ascending
 * index): index of the maximum in the stream (WITHOUT the pretrigger shift of
 * oftrigger.py:1005, which the caller adds), its delta chi2 and its filtered
 * amplitude(s) (amplitude array: max_triggers x n_amp floats, trigger-major).
 * *n_triggers receives the number found (may exceed max_triggers:
 * then only the first max_triggers are written and OFX_ERR_ARG is returned).
 */
int ofx_trigger_find(ofx_trigger* trig, double chi2_threshold, long long pileup_window,
                     long long* index, float* delta_chi2, float* amplitude,
                     long long max_triggers, long long* n_triggers, void* stream);

/*
 * Pieces of find_triggers_once(dynamic=True) and find_triggers(residual=True)
 * (detprocess/core/oftrigger.py:78-143, 752-845, 982-986).
 *
 * ofx_trigger_above: the samples with delta chi2 > chi2_threshold, compacted in ascending order
 * (np.where(triggers_mask)[0] and delta_chi2[triggers_mask], oftrigger.py:976-978) into HOST
 * arrays of capacity cap; *n receives their number (OFX_ERR_ARG if it exceeds cap).  The
 * dynamic pile-up window is a user-supplied Python function of the running range maximum
 * (_getchangeslessthandynamicthresh), so the caller segments this list on the host.
 *
 * ofx_trigger_gather: filtered amplitudes (m x n_amp floats, trigger-major) and delta chi2
 * (may be NULL) at m stream indices (HOST arrays); indices outside the stream give 0.
 *
 * ofx_trigger_set_pulse_table: G[a][b][z], z < n_samples (fp64, host): the delta-chi2 trace
 * of a best-fit pulse with amplitudes A is sum_ab A_a A_b G_ab[z] (oftrigger.py:793-809 with
 * the amplitudes factored out of the convolutions).
 *
 * ofx_trigger_residual_subtract: for every trigger index ti (HOST array; as stored in
 * trigger_index, i.e. with the pretrigger shift, as oftrigger.py:766-815 uses it): A =
 * filtered[:, ti], pulse = sum_ab A_a A_b G_ab, j = first arg-max of pulse,
 * delta_chi2[ti - j : ti - j + n_samples] -= pulse.  The first call after update_trace keeps
 * a copy of the delta-chi2 trace; ofx_trigger_residual_restore copies the residual trace out
 * (pointer may be NULL; mem = OFX_MEM_HOST / OFX_MEM_DEVICE) and puts the first-pass trace back
 * (oftrigger.py:824-828).
 */
int ofx_trigger_above(ofx_trigger* trig, double chi2_threshold, long long* index,
                      float* delta_chi2, long long cap, long long* n, void* stream);
int ofx_trigger_gather(ofx_trigger* trig, const long long* index, long long m, float* amplitude,
                       float* delta_chi2, void* stream);
int ofx_trigger_set_pulse_table(ofx_trigger* trig, const double* G);
int ofx_trigger_residual_subtract(ofx_trigger* trig, const long long* trigger_index, long long m,
                                  void* stream);
int ofx_trigger_residual_restore(ofx_trigger* trig, float* residual_delta_chi2, int mem,
                                 void* stream);

/* ------------------------------------------------------------------------
 * N-channel x M-template optimal filter.
 * Replaces, per batch of events, qp.OFnxm(of_base=, channels='a|b', template_tag=).calc()
 * + get_fit_withdelay(window..., lgc_outside_window) + get_fit_nodelay() as called by
 * FeatureExtractors.ofnxm (detprocess/core/algorithms.py:241-262), and the per-channel
 * update_signal / calc_signal_filt(_td) of processing_data.py:746-772 for those channels.
 * n_chan, n_tmpl <= 4; n_samples even.  The one-time precompute (template FFTs, inverse
 * CSD per bin, weight matrix: processing_data.py:294-381) is done by the caller in fp64.
 * --------------------------------------------------------------------- */
typedef struct ofx_nxm ofx_nxm;

int ofx_nxm_create(ofx_nxm** out, int n_samples, int n_pretrigger, double fs, int n_chan,
                   int n_tmpl, int max_batch, int device);
int ofx_nxm_destroy(ofx_nxm* nxm);

/*
 * One-sided fp64 tables, K = n_samples/2 + 1 bins, interleaved (re, im):
 *   phi  [n_tmpl][n_chan][K]  phi_mb(k) = sum_a conj(S_am(k)) Ci_ab(k)   (NumPy FFT of the
 *                             templates, Ci = inverse two-sided CSD, 0 at dropped bins)
 *   icov [n_chan][n_chan][K]  Ci_ab(k)
 *   pinv [n_tmpl][n_tmpl]     inverse of P_mm' = Re sum_k sum_b phi_mb S_bm' / (N fs)
 */
int ofx_nxm_set_filter(ofx_nxm* nxm, const double* phi, const double* icov, const double* pinv);

/* kind OFX_SEARCH_NODELAY (n = 0), OFX_SEARCH_DELAY over rolled bins [lo, hi) (outside != 0:
 * the complement) or OFX_SEARCH_DELAY_INTERP (the same with interpolate_t0: parabolic refinement
 * of t0, chi2 and every amplitude around the discrete minimum, algorithms.py:152, 259).  Returns the search id (>= 0) or -OFX_ERR_*.  Output record of search s:
 * floats [s (n_tmpl + 3) ...]: amplitudes (n_tmpl), t0 (s), chi2, rolled index. */
int ofx_nxm_add_search(ofx_nxm* nxm, int kind, int lo, int hi, int outside);
int ofx_nxm_reset_searches(ofx_nxm* nxm);

/* events arrive as [n_events][n_channels_total][n_samples]; channel b of the fit is
 * index[b] (default: the first n_chan channels of n_chan) */
int ofx_nxm_set_channels(ofx_nxm* nxm, int n_channels_total, const int* index);
int ofx_nxm_row_floats(const ofx_nxm* nxm);

/* as ofx_process: host buffers are staged in max_batch chunks, device pointers are
 * processed in place on the given hipStream_t; events with valid[e] == 0 give sentinel rows */
int ofx_nxm_process(ofx_nxm* nxm, const float* events, const uint8_t* valid, long long n_events,
                    int events_mem, float* out, int out_mem, void* stream);

/* as ofx_process_adc: the events are cut on the GPU out of continuous int16 streams
 * adc[n_channels_total][n_stream] around trigger_index[e] (window [t - n_pretrigger,
 * t - n_pretrigger + n_samples), amps = float32(adc * scale[c]) + offset[c]; a window that does
 * not fit gives a sentinel row: processing_data.py:640-656, 674-684), then fitted as above.
 * trigger_index, scale, offset and (for OFX_MEM_HOST) out are host arrays. */
int ofx_nxm_process_adc(ofx_nxm* nxm, const int16_t* adc, long long n_stream, int adc_mem,
                        const long long* trigger_index, long long n_events, const double* scale,
                        const double* offset, float* out, int out_mem, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OFX_H */
