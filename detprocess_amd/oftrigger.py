"""OptimumFilterTrigger: continuous-data optimal-filter trigger on the GPU -- the interface of
detprocess/core/oftrigger.py:336-1035, N channels x M amplitudes.

The one-time filter precompute (oftrigger.py:466-496, QETpy ``OFBase`` phi / weight /
iweight) stays on the host in fp64; ``update_trace`` (FIR filtering of the whole stream, delta
chi2, edge padding), the threshold / range-merging / arg-max core of ``find_triggers_once``
and the pulse subtraction of the residual pass run in HIP through ``ofx_trigger_*``
(include/ofx.h).  There is no CPU fallback.

Two steps are driven from the host because their inputs are host objects: the dynamic
pile-up window is a user-supplied Python function of the running range maximum
(oftrigger.py:78-143), so the device compacts the above-threshold samples and the ranges are
cut here; and the saturation veto of the residual pass (oftrigger.py:772-786) low-passes
the raw trace around each first-pass trigger only (a few hundred windows, not the stream).
"""

import ctypes as C
import warnings

import numpy as np

from . import _lib
from .filters import apply_coupling_and_notches, build_filter


def _dynamic_ranges(x, amplitudes, threshold_function):
    """Ranges [start, end) of the above-threshold samples x (ascending stream indices) for a
    pile-up window that depends on the largest delta chi2 seen so far in the current range
    (oftrigger.py:78-143): sample i opens a new range when x[i] - x[i-1] exceeds
    threshold_function(max(amplitudes[start .. i])).  One pass with a running maximum; the
    function is called again only when that maximum has changed."""
    n = len(x)
    if n == 0:
        return []
    cuts = [0]
    run_max, window = amplitudes[0], None
    for i in range(1, n):
        if amplitudes[i] > run_max:
            run_max, window = amplitudes[i], None
        if window is None:
            window = threshold_function(run_max)
        if (x[i] - x[i - 1]) > window:
            cuts.append(i)
            run_max, window = amplitudes[i], None
    return list(zip(cuts, cuts[1:] + [n]))


def _lowpass_50khz_window(raw, fs, start, stop, pad=1024):
    """qp.utils.lowpassfilter(trace, cut_off_freq=50e3, fs) -- first-order Butterworth run
    forward and backward (filtfilt, even padding) -- evaluated on raw[start:stop] from a
    window extended by ``pad`` samples on both sides: the filter's memory (pole 0.78 at 50 kHz /
    1.25 MHz) is gone after a few dozen samples, so this equals the whole-stream result of
    oftrigger.py:627-633 to rounding, at the cost of the windows actually looked at."""
    from scipy.signal import butter, filtfilt
    n = raw.shape[-1]
    a0, a1 = max(0, start - pad), min(n, stop + pad)
    seg = np.asarray(raw[a0:a1], dtype=np.float64)
    b, a = butter(1, 50e3 / (0.5 * fs))
    if seg.shape[-1] <= 9:
        return seg[start - a0: stop - a0]
    return filtfilt(b, a, seg, padtype="even")[start - a0: stop - a0]


def pulse_table(template, phi_td, iw_eff, w_matrix):
    """G[a][b][z]: the delta-chi2 trace of a best-fit pulse with amplitudes A is
    sum_ab A_a A_b G_ab[z] -- oftrigger.py:793-809 with the amplitudes taken out of the
    convolutions.  template [C, M, T], phi_td [C, M, T] (as the reference stores them), iw_eff the
    matrix that turns the summed convolutions into amplitudes, w_matrix the weight matrix.  The
    filter of amplitude theta is indexed as the reference indexes it in this loop
    (``self._phi_td[theta, :]``, oftrigger.py:800; update_trace uses ``[:, theta, :]``, :658):
    identical for one channel x one amplitude, and the reference's own expression only runs when
    the two counts agree (or one is 1); other shapes use the update_trace convention."""
    from scipy.signal import oaconvolve
    C_, M, T = template.shape
    lit = (C_ == M) or C_ == 1 or M == 1
    f = np.zeros((M, M, T))                            # f[a][i][z] = (iw U_a)[i][z]
    for a in range(M):
        U = np.zeros((M, T))
        for theta in range(M):
            kern = phi_td[theta] if lit else phi_td[:, theta, :]
            kern = np.broadcast_to(kern, (C_, T)) if kern.shape[0] != C_ else kern
            U[theta] = np.sum(oaconvolve(template[:, a, :], kern, mode="same", axes=-1), axis=0)
        f[a] = iw_eff @ U
    return np.ascontiguousarray(np.einsum("aiz,ij,bjz->abz", f, w_matrix, f))


def _chi2_threshold(thresh, m_amplitudes=1):
    """sigma -> chi2 threshold, oftrigger.py:962-975."""
    from scipy import special, stats
    if thresh < 25:
        survival_fraction = stats.norm.sf(thresh) * 2
        return float(special.gammainccinv(m_amplitudes / 2, survival_fraction) * 2)
    if m_amplitudes > 1:
        warnings.warn("threshold too high for the chi2 conversion; using the M = 1 result")
    return float(thresh) ** 2


class OptimumFilterTrigger:
    def __init__(self, trigger_channel, fs, template, noisecsd, pretrigger_samples,
                 trigger_name=None, ignored_frequency_peaks=None, ignore_harmonics=False,
                 device=0):
        template = np.asarray(template, dtype=np.float64)
        noisecsd = np.asarray(noisecsd)
        # shapes as oftrigger.py:409-441: template -> [channels, amplitudes, samples],
        # csd -> [channels, channels, frequencies]
        if template.ndim == 1:
            template = template.reshape(1, 1, -1)
        elif template.ndim == 2:
            if 1 not in template.shape:
                raise ValueError(f"Template is shaped as {template.shape}. It should be (N, M, "
                                 "samples) or (samples,) or (1, samples) or (samples, 1).")
            template = template.reshape(1, 1, -1)
        elif template.ndim != 3:
            raise ValueError(f"Template is shaped as {template.shape}. It should be (N, M, "
                             "samples) or (samples,) or (1, samples) or (samples, 1).")
        if noisecsd.ndim == 1:
            noisecsd = noisecsd.reshape(1, 1, -1)
        elif noisecsd.ndim == 2:
            if 1 not in noisecsd.shape:
                raise ValueError(f"Noise CSD is shaped as {noisecsd.shape}. Should be (N, M, "
                                 "frequencies) or (frequencies,) or (1, frequencies) or "
                                 "(frequencies, 1).")
            noisecsd = noisecsd.reshape(1, 1, -1)
        elif noisecsd.ndim != 3:
            raise ValueError(f"Noise CSD is shaped as {noisecsd.shape}. Should be (N, M, "
                             "frequencies) or (frequencies,) or (1, frequencies) or "
                             "(frequencies, 1).")
        self._fs = float(fs)
        self._pretrigger_samples = int(pretrigger_samples)
        self._trigger_channel = (trigger_channel if isinstance(trigger_channel, str)
                                 else "|".join(trigger_channel))
        self._trigger_name = str(trigger_name) if trigger_name is not None else \
            str(self._trigger_channel)
        self._trigger_name = self._trigger_name.replace("\0", "")
        self._template = template
        self._nb_samples = template.shape[-1]
        self._posttrigger_samples = self._nb_samples - self._pretrigger_samples
        self._n_channels, _, self._f_frequencies = noisecsd.shape
        self._m_amplitudes = template.shape[1]
        if template.shape[0] != self._n_channels or noisecsd.shape[1] != self._n_channels:
            raise ValueError(f"ERROR: template {template.shape} and csd {noisecsd.shape} do not "
                             "describe the same channels")
        self._t_times = self._nb_samples
        self._trigger_index_shift = self._pretrigger_samples - self._nb_samples // 2
        self._lib = _lib.load()
        self._h = C.c_void_p()
        n = self._nb_samples
        if self._n_channels == 1 and self._m_amplitudes == 1:
            # filter precompute (host, fp64)
            t1 = template[0, 0]
            psd = np.real(noisecsd[0, 0]).astype(np.float64)
            tables = build_filter(t1, psd, self._fs, self._pretrigger_samples, "AC",
                                  ignored_frequency_peaks, ignore_harmonics)
            J = apply_coupling_and_notches(psd, self._fs, "AC", ignored_frequency_peaks,
                                           ignore_harmonics)
            S = np.fft.fft(t1)
            with np.errstate(divide="ignore"):
                phi_fd = np.where(np.isfinite(J), np.conj(S) / J, 0.0)
            phi_fd[0] = 0.0                                    # oftrigger.py:488
            self._phi_td = np.ascontiguousarray(np.fft.ifft(phi_fd).real).reshape(1, 1, -1)
            self._w_matrix = np.array([[float(tables.norm)]])
            self._iw_matrix = np.array([[1.0 / float(tables.norm)]])
            self._resolution = np.array([tables.ampres])
            _lib.check(self._lib.ofx_trigger_create(
                C.byref(self._h), int(n), int(self._pretrigger_samples), self._fs,
                self._phi_td.ctypes.data, float(tables.norm) * self._fs,
                float(tables.norm), int(device)), "ofx_trigger_create")
        else:
            if n % 2:
                raise ValueError("ERROR: the NxM trigger needs an even number of samples")
            from .ofnxm import build_nxm_filter
            tab = build_nxm_filter(template, noisecsd, self._fs, self._pretrigger_samples, "AC",
                                   ignored_frequency_peaks, ignore_harmonics)
            phi = tab.phi.copy()                               # [m, b, k] one-sided
            phi[:, :, 0] = 0.0                                 # oftrigger.py:488
            # ifft(phi).real of the Hermitian spectrum = irfft; stored [channel, amplitude, t]
            phi_td = np.fft.irfft(phi, n=n, axis=-1)
            self._phi_td = np.ascontiguousarray(np.transpose(phi_td, (1, 0, 2)))
            pinv = np.asarray(tab.pinv)
            self._iw_matrix = pinv
            self._w_matrix = np.linalg.inv(pinv)
            self._resolution = np.sqrt(np.diag(pinv))          # oftrigger.py:496
            # conv(x, phi_td)(t) = fs q(t) with q as in ofnxm.py, so amplitudes = P^-1 V_td / fs
            iw_dev = np.ascontiguousarray(pinv / self._fs)
            w_dev = np.ascontiguousarray(self._w_matrix)
            _lib.check(self._lib.ofx_trigger_create_nxm(
                C.byref(self._h), int(n), int(self._pretrigger_samples), self._fs,
                int(self._n_channels), int(self._m_amplitudes), self._phi_td.ctypes.data,
                iw_dev.ctypes.data, w_dev.ctypes.data, int(device)), "ofx_trigger_create_nxm")
        self._norm = float(np.dot(self._phi_td[0, 0], template[0, 0]))      # oftrigger.py:493
        self._device = int(device)
        self._n = 0
        self._trigger_data = None
        self.chi2_threshold = None

    # ------------------------------------------------------------ accessors
    def get_phi(self):
        return self._phi_td

    def get_norm(self):
        return self._norm

    def get_resolution(self):
        return self._resolution

    def get_chi2_threshold(self):
        return self.chi2_threshold

    def get_trigger_data(self):
        return self._trigger_data

    def _traces(self):
        if self._n == 0:
            return None, None
        f = np.empty((self._m_amplitudes, self._n), dtype=np.float32)
        d = np.empty(self._n, dtype=np.float32)
        _lib.check(self._lib.ofx_trigger_get_traces(self._h, f.ctypes.data, d.ctypes.data,
                                                    _lib.MEM_HOST, None), "ofx_trigger_get_traces")
        return f, d

    def get_filtered_trace(self):
        return self._traces()[0]

    def get_filtered_delta_chi2(self):
        return self._traces()[1]

    # ----------------------------------------------------------- update_trace
    def update_trace(self, trace=None, filtered_trace=None, padding=True, adc_scale=None,
                     adc_offset=0.0):
        """oftrigger.py:588-679.  trace: [samples] or [1, samples]; float (amps) NumPy array
        or CUDA tensor, or int16 raw ADC values with ``adc_scale`` / ``adc_offset``
        (amps = adc * scale + offset)."""
        if filtered_trace is not None:
            raise NotImplementedError("a pre-filtered trace is not accepted by the GPU trigger")
        if trace is None:
            raise ValueError('ERROR: "trace" or "filtered_trace required!')
        is_np = isinstance(trace, np.ndarray)
        nc = self._n_channels
        if trace.ndim == 1:
            trace = trace.reshape(1, -1)
        if trace.ndim != 2 or trace.shape[0] != nc:
            raise ValueError(f'ERROR: "trace" has shape {tuple(trace.shape)}, but we have '
                             f"{nc} channels!")
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(
            1.0 if adc_scale is None else adc_scale, dtype=np.float64), (nc,)))
        of = np.ascontiguousarray(np.broadcast_to(np.asarray(adc_offset, dtype=np.float64), (nc,)))
        if is_np:
            if trace.dtype == np.int16:
                if adc_scale is None:
                    raise ValueError("ERROR: int16 input needs adc_scale")
                x = np.ascontiguousarray(trace)
                dtype, ptr = 1, x.ctypes.data
            else:
                x = np.ascontiguousarray(trace, dtype=np.float32)
                dtype, ptr = 0, x.ctypes.data
            mem, stream = _lib.MEM_HOST, None
        else:
            import torch
            if not trace.is_cuda:
                return self.update_trace(trace.numpy(), padding=padding, adc_scale=adc_scale,
                                         adc_offset=adc_offset)
            x = trace.contiguous()
            if x.dtype == torch.int16:
                if adc_scale is None:
                    raise ValueError("ERROR: int16 input needs adc_scale")
                dtype = 1
            else:
                x = x.to(torch.float32)
                dtype = 0
            ptr, mem = x.data_ptr(), _lib.MEM_DEVICE
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        n = int(x.shape[-1])
        _lib.check(self._lib.ofx_trigger_update_traces(
            self._h, ptr, dtype, n, mem, sc.ctypes.data, of.ctypes.data, int(bool(padding)),
            stream), "ofx_trigger_update_traces")
        if not is_np:
            import torch
            torch.cuda.current_stream().synchronize()       # x must outlive the kernels
        self._n = n
        self._raw = (x, dtype, sc, of)                      # saturation veto of the residual pass

    # ---------------------------------------------------------- find_triggers
    def find_triggers(self, thresh, pileup_window_msec=None, pileup_window_samples=None,
                      positive_pulses=True, dynamic=False, dynamic_threshold_function=None,
                      residual=False, saturation_amplitudes_LPF_50kHz=None,
                      edge_exclusion_msec=None, livetime=None, return_trigger_data=False):
        """oftrigger.py:681-880."""
        ret = None
        if residual:
            ret = self._find_triggers_residual(
                thresh, pileup_window_msec, pileup_window_samples, positive_pulses, dynamic,
                dynamic_threshold_function, saturation_amplitudes_LPF_50kHz, return_trigger_data)
        else:
            self.find_triggers_once(thresh, pileup_window_msec, pileup_window_samples, dynamic,
                                    dynamic_threshold_function)
        if edge_exclusion_msec is not None:
            tmin = edge_exclusion_msec * 1e-3
            tmax = self._n / self._fs - edge_exclusion_msec * 1e-3
            for chan in list(self._trigger_data):
                data = self._trigger_data[chan]
                times = data["trigger_time"]
                if len(times) == 0:
                    continue
                keep = [i for i, t in enumerate(times) if tmin < t < tmax]
                out = {k: [v[i] for i in keep] for k, v in data.items()}
                out[f"trigger_edge_exclusion_time_{chan}"] = [edge_exclusion_msec * 1e-3] * len(keep)
                if livetime is not None:
                    out[f"trigger_livetime_{chan}"] = [livetime] * len(keep)
                self._trigger_data[chan] = out
        return ret

    # ------------------------------------------------------------ residual pass
    def _pulse_table(self):
        if self._n_channels == 1 and self._m_amplitudes == 1:
            iw_eff = np.array([[1.0 / (float(self._w_matrix[0, 0]) * self._fs)]])
        else:
            iw_eff = self._iw_matrix / self._fs
        return pulse_table(self._template, self._phi_td, iw_eff, self._w_matrix)

    def _saturated(self, trigger_index, positive_pulses, sat):
        """oftrigger.py:772-786: a first-pass trigger is vetoed when the 50 kHz low-passed raw
        trace of any channel crosses its saturation amplitude within n_samples/4 of it."""
        if all(not np.isfinite(v) for v in sat):
            return False
        x, dtype, sc, of = self._raw
        q = int(self._t_times / 4)
        lo, hi = trigger_index - q, trigger_index + q
        if lo < 0 or hi <= lo:
            return False
        for ch in range(self._n_channels):
            if not np.isfinite(sat[ch]):
                continue
            row = x[ch]
            a0, a1 = max(0, lo - 1024), min(row.shape[-1], hi + 1024)
            seg = row[a0:a1]
            seg = seg.cpu().numpy() if not isinstance(seg, np.ndarray) else seg
            seg = seg.astype(np.float64)
            if dtype == 1:
                seg = seg * sc[ch] + of[ch]
            lp = _lowpass_50khz_window(seg, self._fs, lo - a0, min(hi, row.shape[-1]) - a0, pad=0)
            if positive_pulses:
                if np.sum(lp > sat[ch]) > 0:
                    return True
            elif np.sum(lp < -1 * sat[ch]) > 0:
                return True
        return False

    def _find_triggers_residual(self, thresh, pileup_window_msec, pileup_window_samples,
                                positive_pulses, dynamic, dynamic_threshold_function, sat,
                                return_trigger_data):
        import copy
        if sat is None:
            sat = [np.inf if positive_pulses else -np.inf] * self._n_channels
        self.find_triggers_once(thresh, pileup_window_msec, pileup_window_samples, dynamic,
                                dynamic_threshold_function)
        name = self._trigger_name
        original_triggers = list(self._trigger_data[name]["trigger_index"])
        original_trigger_data = copy.deepcopy(self._trigger_data)
        if not getattr(self, "_pulse_set", False):
            G = self._pulse_table()
            _lib.check(self._lib.ofx_trigger_set_pulse_table(self._h, G.ctypes.data),
                       "ofx_trigger_set_pulse_table")
            self._pulse_set = True
        keep = np.asarray([ti for ti in original_triggers
                           if not self._saturated(int(ti), positive_pulses, sat)], dtype=np.int64)
        res = np.empty(self._n, dtype=np.float32) if return_trigger_data else None
        try:
            # (inside the try: a failure after the first-pass trace was saved, or half-way through the
            # subtraction, must still put it back)
            _lib.check(self._lib.ofx_trigger_residual_subtract(
                self._h, keep.ctypes.data if len(keep) else None, len(keep), None),
                "ofx_trigger_residual_subtract")
            self.find_triggers_once(thresh, pileup_window_msec, pileup_window_samples, dynamic,
                                    dynamic_threshold_function)
            new_triggers = list(self._trigger_data[name]["trigger_index"])
            new_trigger_data = copy.deepcopy(self._trigger_data)
        except BaseException:
            # the first-pass trace goes back whatever happened (oftrigger.py:824-828); the residual
            # trace is not asked for on this path, so a subtract that failed before saving is fine
            self._lib.ofx_trigger_residual_restore(self._h, None, _lib.MEM_HOST, None)
            raise
        else:
            _lib.check(self._lib.ofx_trigger_residual_restore(
                self._h, res.ctypes.data if res is not None else None, _lib.MEM_HOST, None),
                "ofx_trigger_residual_restore")
        self._residual_delta_chi2_trace = res
        # combine_trigger_data (oftrigger.py:262-320): second-pass triggers whose index is new
        # are appended; the "<key>_<name>" entries are the same list objects as "<key>"
        fresh = set(new_triggers) - set(original_triggers)
        combined = copy.deepcopy(original_trigger_data[name])
        newd = new_trigger_data[name]
        for key in newd:
            if ("_" + name) in key:
                continue
            if key not in combined:            # (first pass found nothing: no trigger_channel)
                combined[key] = []
                combined[key + "_" + name] = combined[key]
            for i, trig in enumerate(new_triggers):
                if trig in fresh:
                    combined[key].append(newd[key][i])
        self._trigger_data = {name: combined}
        if return_trigger_data:
            return original_trigger_data, self.get_filtered_delta_chi2(), new_trigger_data, res
        return None

    def find_triggers_once(self, thresh, pileup_window_msec=None, pileup_window_samples=None,
                           dynamic=False, dynamic_threshold_function=None):
        """oftrigger.py:884-1035."""
        if self._n == 0:
            raise ValueError('ERROR: Filter trace not available.  Use "update_trace" first!')
        pileup_window = 0
        if pileup_window_msec is not None:
            pileup_window = int(pileup_window_msec * self._fs / 1000)
        elif pileup_window_samples is not None:
            pileup_window = int(pileup_window_samples)
        chi2_threshold = _chi2_threshold(thresh, self._m_amplitudes)
        self.chi2_threshold = chi2_threshold
        if dynamic:
            if dynamic_threshold_function is None:
                raise ValueError('ERROR: "dynamic_threshold_function" required when dynamic=True')
            idx_out, dchi_out, amp_out = self._find_dynamic(chi2_threshold,
                                                            dynamic_threshold_function)
        else:
            idx_out, dchi_out, amp_out = self._find_static(chi2_threshold, pileup_window)
        m = len(idx_out)
        ind = idx_out + self._trigger_index_shift                       # oftrigger.py:1005
        data = {
            "trigger_delta_chi2": [float(v) for v in dchi_out],
            "trigger_time": [float(v) for v in ind / self._fs],
            "trigger_index": [int(v) for v in ind],
            "trigger_pileup_window": [pileup_window] * m,
            "trigger_threshold_sigma": [thresh] * m,
            "trigger_type": [4] * m,
        }
        for iamp in range(self._m_amplitudes):                          # oftrigger.py:926-929
            data[f"trigger_amplitude_{iamp}"] = [float(v) for v in amp_out[:, iamp]]
        if self._m_amplitudes == 1:
            data["trigger_amplitude"] = [float(v) for v in amp_out[:, 0]]
        if m > 0:
            data["trigger_channel"] = [str(self._trigger_name)] * m
        self._trigger_data = {self._trigger_name: dict(data)}
        for key, val in data.items():                                   # oftrigger.py:1031-1033
            self._trigger_data[self._trigger_name][key + "_" + self._trigger_name] = val

    def _find_static(self, chi2_threshold, pileup_window):
        cap = 1 << 16
        while True:
            idx = np.empty(cap, dtype=np.int64)
            dchi = np.empty(cap, dtype=np.float32)
            amp = np.empty((cap, self._m_amplitudes), dtype=np.float32)
            cnt = C.c_longlong()
            rc = self._lib.ofx_trigger_find(self._h, chi2_threshold, pileup_window,
                                            idx.ctypes.data, dchi.ctypes.data, amp.ctypes.data,
                                            cap, C.byref(cnt), None)
            if rc == 0:
                break
            if cnt.value > cap:
                cap = int(cnt.value)
                continue
            _lib.check(rc, "ofx_trigger_find")
        m = cnt.value
        return idx[:m], dchi[:m], amp[:m]

    def _find_dynamic(self, chi2_threshold, threshold_function):
        """oftrigger.py:982-986 + 993-1019: the device compacts the samples above threshold;
        the ranges (a function of the running maximum through a Python callable) are cut here;
        the first maximum of each range is the trigger."""
        cap = 1 << 20
        while True:
            idx = np.empty(cap, dtype=np.int64)
            dchi = np.empty(cap, dtype=np.float32)
            cnt = C.c_longlong()
            rc = self._lib.ofx_trigger_above(self._h, chi2_threshold, idx.ctypes.data,
                                             dchi.ctypes.data, cap, C.byref(cnt), None)
            if rc == 0:
                break
            if cnt.value > cap:
                cap = int(cnt.value)
                continue
            _lib.check(rc, "ofx_trigger_above")
        n = cnt.value
        idx, dchi = idx[:n], dchi[:n]
        picks = np.asarray([s + int(np.argmax(dchi[s:e]))
                            for s, e in _dynamic_ranges(idx, dchi, threshold_function)],
                           dtype=np.int64)
        sel = np.ascontiguousarray(idx[picks]) if len(picks) else np.empty(0, dtype=np.int64)
        amp = np.empty((len(sel), self._m_amplitudes), dtype=np.float32)
        if len(sel):
            _lib.check(self._lib.ofx_trigger_gather(self._h, sel.ctypes.data, len(sel),
                                                    amp.ctypes.data, None, None),
                       "ofx_trigger_gather")
        return sel, dchi[picks] if len(picks) else np.empty(0, dtype=np.float32), amp

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ofx_trigger_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
