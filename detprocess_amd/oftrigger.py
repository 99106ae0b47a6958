"""OptimumFilterTrigger: continuous-data optimal-filter trigger on the GPU, one channel x one
amplitude -- the interface of detprocess/core/oftrigger.py:336-1035 for that case
(SURVEY.md section 8f rank 3).

The one-time filter precompute (oftrigger.py:466-496, QETpy ``OFBase`` phi / weight /
iweight) stays on the host in fp64; ``update_trace`` (FIR filtering of the whole stream, delta
chi2, edge padding) and the threshold / range-merging / arg-max core of ``find_triggers_once``
run in HIP through ``ofx_trigger_*`` (include/ofx.h).  There is no CPU fallback.
"""

import ctypes as C
import warnings

import numpy as np

from . import _lib
from .filters import apply_coupling_and_notches, build_filter


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

    # ---------------------------------------------------------- find_triggers
    def find_triggers(self, thresh, pileup_window_msec=None, pileup_window_samples=None,
                      positive_pulses=True, dynamic=False, dynamic_threshold_function=None,
                      residual=False, saturation_amplitudes_LPF_50kHz=None,
                      edge_exclusion_msec=None, livetime=None, return_trigger_data=False):
        """oftrigger.py:681-880 without the residual pass."""
        if residual:
            raise NotImplementedError("the residual re-trigger pass is not on the GPU")
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

    def find_triggers_once(self, thresh, pileup_window_msec=None, pileup_window_samples=None,
                           dynamic=False, dynamic_threshold_function=None):
        """oftrigger.py:884-1035 (static pile-up window)."""
        if self._n == 0:
            raise ValueError('ERROR: Filter trace not available.  Use "update_trace" first!')
        if dynamic:
            raise NotImplementedError("the dynamic pile-up window is not on the GPU")
        pileup_window = 0
        if pileup_window_msec is not None:
            pileup_window = int(pileup_window_msec * self._fs / 1000)
        elif pileup_window_samples is not None:
            pileup_window = int(pileup_window_samples)
        chi2_threshold = _chi2_threshold(thresh, self._m_amplitudes)
        self.chi2_threshold = chi2_threshold
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
        ind = idx[:m] + self._trigger_index_shift                       # oftrigger.py:1005
        data = {
            "trigger_delta_chi2": [float(v) for v in dchi[:m]],
            "trigger_time": [float(v) for v in ind / self._fs],
            "trigger_index": [int(v) for v in ind],
            "trigger_pileup_window": [pileup_window] * m,
            "trigger_threshold_sigma": [thresh] * m,
            "trigger_type": [4] * m,
        }
        for iamp in range(self._m_amplitudes):                          # oftrigger.py:926-929
            data[f"trigger_amplitude_{iamp}"] = [float(v) for v in amp[:m, iamp]]
        if self._m_amplitudes == 1:
            data["trigger_amplitude"] = [float(v) for v in amp[:m, 0]]
        if m > 0:
            data["trigger_channel"] = [str(self._trigger_name)] * m
        self._trigger_data = {self._trigger_name: dict(data)}
        for key, val in data.items():                                   # oftrigger.py:1031-1033
            self._trigger_data[self._trigger_name][key + "_" + self._trigger_name] = val

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ofx_trigger_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
