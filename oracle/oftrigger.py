"""TEST INFRASTRUCTURE ONLY -- CPU restatement (fp64 NumPy / SciPy) of the continuous-data
optimal-filter trigger of detprocess (``OFTrigger``: single channel x single amplitude;
``OFTriggerNxM``: N channels x M amplitudes):

    OptimumFilterTrigger.__init__          detprocess/core/oftrigger.py:384-499
    OptimumFilterTrigger.update_trace      detprocess/core/oftrigger.py:588-679
    OptimumFilterTrigger.find_triggers_once  :884-1035   (static and dynamic pile-up window)
    _getchangeslessthanthresh              :29-77
    _getchangeslessthandynamicthresh       :78-143
    find_triggers(residual=True)           :752-845  (saturation veto, pulse subtraction in
                                            delta-chi2 space, re-trigger, combine_trigger_data
                                            :262-320)
    edge exclusion of find_triggers        :851-880
qp.utils.lowpassfilter (oftrigger.py:629-633; QETpy, unseen) is restated as a first-order
Butterworth run through scipy.signal.filtfilt with even padding.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

PARITY UNPINNED for the part that lives in QETpy (qp.OFBase.phi / weight / iweight, absent
from /root/reference): restated here as phi_k = conj(S_k) / J_k (S = fft(template), J the
two-sided PSD, infinite at DC and at ignored peaks), w = norm of oracle/of1x1.py (1 / ampres^2),
and the product iw x (scale of phi) fixed by the one property the estimator must have: the
filtered trace of A x template equals A at the pulse, i.e. filtered = conv(trace, phi_td) /
((1/N) sum_k |S_k|^2 / J_k) = conv / (norm fs).  Everything after the filter -- 'same'-mode
overlap-add convolution, edge padding, the sigma -> chi2 threshold, range merging, arg-max,
index shift -- is in the reference tree and is followed line by line.  With phi_td =
ifft(conj(S)/J) the in-tree arithmetic puts the maximum one sample after onset + pretrigger
(N even): conv_full peaks at onset + N, 'same' removes (N-1)//2, the shift adds
pretrigger - N//2; whether QETpy's phi compensates for it cannot be seen from here.
"""
import numpy as np
from scipy import special, stats
from scipy.signal import oaconvolve

from . import of1x1


def dynamic_ranges(x, amplitudes, threshold_function):
    """_getchangeslessthandynamicthresh, oftrigger.py:111-135, literally: the maximum of the
    current range is recomputed from its start at every step."""
    starts, ends, cur = [], [], 0
    for i in range(1, len(x)):
        if (x[i] - x[i - 1]) > threshold_function(np.max(amplitudes[cur:i + 1])):
            starts.append(cur)
            ends.append(i)
            cur = i
    starts.append(cur)
    ends.append(len(x))
    return list(zip(starts, ends))


def lowpass_50khz(trace, fs):
    """qp.utils.lowpassfilter(trace, cut_off_freq=50e3, fs=fs): butter(1) + filtfilt(even)."""
    from scipy.signal import butter, filtfilt
    b, a = butter(1, 50e3 / (0.5 * fs))
    return filtfilt(b, a, np.asarray(trace, dtype=np.float64), padtype="even")


def _ranges(trig, dchi, window, dynamic_function):
    if dynamic_function is not None:
        return dynamic_ranges(trig, dchi[trig], dynamic_function) if len(trig) else []
    if not len(trig):
        return []
    cuts = np.where((trig[1:] - trig[:-1]) > window)[0] + 1
    return list(zip(np.concatenate(([0], cuts)), np.concatenate((cuts, [len(trig)]))))


def residual_pass(obj, find_once, trigger_index, positive_pulses, saturation, raw, pulse_of):
    """find_triggers(residual=True), oftrigger.py:752-845, shared by the two oracle classes:
    returns (first-pass result, second-pass result, residual delta-chi2 trace, combined index
    list in the order combine_trigger_data produces)."""
    first = find_once()
    original = np.copy(obj.delta_chi2)
    T = obj.N
    lp = None
    for ti in first["trigger_index"]:
        saturated = False
        for ch in range(raw.shape[0]):
            if not np.isfinite(saturation[ch]):
                continue
            if lp is None:
                lp = np.stack([lowpass_50khz(r, obj.fs) for r in raw])
            seg = lp[ch][ti - int(T / 4): ti + int(T / 4)]
            if positive_pulses:
                saturated |= bool(np.sum(seg > saturation[ch]) > 0)
            else:
                saturated |= bool(np.sum(seg < -1 * saturation[ch]) > 0)
        if saturated:
            continue
        pulse = pulse_of(ti)                        # delta chi2 of the best-fit pulse, length T
        j = int(np.argmax(pulse))
        obj.delta_chi2[ti - j: ti - j + T] -= pulse
    second = find_once()
    residual = np.copy(obj.delta_chi2)
    obj.delta_chi2 = original
    fresh = set(second["trigger_index"].tolist()) - set(first["trigger_index"].tolist())
    combined = list(first["trigger_index"]) + [t for t in second["trigger_index"] if t in fresh]
    return first, second, residual, np.asarray(combined, dtype=np.int64)


class OFTrigger:
    def __init__(self, fs, template, psd, pretrigger_samples, coupling="AC",
                 ignored_frequency_peaks=None, ignore_harmonics=False):
        self.fs = float(fs)
        self.template = np.asarray(template, dtype=np.float64)
        self.N = self.template.shape[-1]
        self.pre = int(pretrigger_samples)
        J = of1x1.effective_psd(np.asarray(psd, dtype=np.float64), self.fs, coupling,
                                ignored_frequency_peaks, ignore_harmonics)
        S = np.fft.fft(self.template)
        with np.errstate(divide="ignore"):
            phi_fd = np.conj(S) / J
        phi_fd[~np.isfinite(J)] = 0.0
        phi_fd[0] = 0.0                                  # oftrigger.py:488 no DC information
        self.phi_td = np.fft.ifft(phi_fd).real           # oftrigger.py:489
        self.norm_td = float(np.dot(self.phi_td, self.template))      # oftrigger.py:493 (get_norm)
        filt = of1x1.OFFilter(self.template, psd, self.fs, self.pre, coupling,
                              ignored_frequency_peaks, ignore_harmonics)
        self.w = float(filt.norm)                        # weight matrix (1x1)
        self.vscale = float(filt.norm) * self.fs         # (1/N) sum |S|^2 / J
        self.resolution = float(filt.ampres)             # sqrt(diag(iweight)), oftrigger.py:496
        self.index_shift = self.pre - self.N // 2        # oftrigger.py:455
        self.filtered = None
        self.delta_chi2 = None

    def update_trace(self, trace, padding=True):
        """oftrigger.py:649-679 for n_channels = m_amplitudes = 1."""
        x = np.asarray(trace, dtype=np.float64).reshape(-1)
        v = oaconvolve(x, self.phi_td, mode="same")
        self.filtered = v / self.vscale
        self.delta_chi2 = self.filtered * self.w * self.filtered
        if padding:
            cut = self.N
            self.delta_chi2[:cut] = 0.0
            self.delta_chi2[-(cut) + (cut + 1) % 2:] = 0.0
        return self.filtered, self.delta_chi2

    @staticmethod
    def chi2_threshold(thresh, m_amplitudes=1):
        """oftrigger.py:962-975."""
        if thresh < 25:
            sf = stats.norm.sf(thresh) * 2
            return float(special.gammainccinv(m_amplitudes / 2, sf) * 2)
        return float(thresh) ** 2

    def find_triggers_residual(self, thresh, raw, pileup_window_samples=None,
                               dynamic_function=None, positive_pulses=True, saturation=None):
        """oftrigger.py:752-845 for one channel x one amplitude."""
        from scipy.signal import oaconvolve as oac
        raw = np.asarray(raw, dtype=np.float64).reshape(1, -1)
        if saturation is None:
            saturation = [np.inf if positive_pulses else -np.inf]

        def pulse_of(ti):
            amp = self.filtered[ti]                               # oftrigger.py:792
            trig_trace = self.template * amp
            v = oac(trig_trace, self.phi_td, mode="same")
            filt = v / self.vscale
            return filt * self.w * filt

        once = lambda: self.find_triggers(thresh, pileup_window_samples=pileup_window_samples,
                                          dynamic_function=dynamic_function)
        return residual_pass(self, once, None, positive_pulses, saturation, raw, pulse_of)

    def find_triggers(self, thresh, pileup_window_msec=None, pileup_window_samples=None,
                      edge_exclusion_msec=None, dynamic_function=None):
        """find_triggers_once (static or dynamic window) + the edge exclusion of find_triggers.
        Returns dict of arrays: trigger_index, trigger_time, trigger_delta_chi2,
        trigger_amplitude."""
        window = 0
        if pileup_window_msec is not None:
            window = int(pileup_window_msec * self.fs / 1000)
        elif pileup_window_samples is not None:
            window = pileup_window_samples
        thr = self.chi2_threshold(thresh)
        mask = self.delta_chi2 > thr
        trig = np.where(mask)[0]
        idx, dchi, amp = [], [], []
        # _getchangeslessthanthresh: split where consecutive indices differ by > window;
        # _getchangeslessthandynamicthresh when a window function is given
        for s, e in _ranges(trig, self.delta_chi2, window, dynamic_function):
            if e > s:
                inds = trig[s:e]
                i = inds[np.argmax(self.delta_chi2[inds])]
                idx.append(i + self.index_shift)
                dchi.append(self.delta_chi2[i])
                amp.append(self.filtered[i])
        idx = np.asarray(idx, dtype=np.int64)
        out = {"trigger_index": idx, "trigger_time": idx / self.fs,
               "trigger_delta_chi2": np.asarray(dchi, dtype=np.float64),
               "trigger_amplitude": np.asarray(amp, dtype=np.float64),
               "chi2_threshold": thr, "pileup_window": window}
        if edge_exclusion_msec is not None:
            tmin = edge_exclusion_msec * 1e-3
            tmax = self.filtered.shape[-1] / self.fs - edge_exclusion_msec * 1e-3
            keep = (out["trigger_time"] > tmin) & (out["trigger_time"] < tmax)
            for k in ("trigger_index", "trigger_time", "trigger_delta_chi2", "trigger_amplitude"):
                out[k] = out[k][keep]
        return out


class OFTriggerNxM:
    """N channels x M amplitudes (oftrigger.py:407-499, 649-679, 926-1019).  phi, weight and
    iweight are QETpy's (unpinned, restated as in oracle/ofnxm.py): phi_bm(k) = sum_a conj(S_am)
    Ci_ab with the DC bin zeroed, weight = P, and iweight x (scale of phi) = P^-1 / fs so that
    the filtered trace of sum_m A_m template_m equals A at the pulse."""

    def __init__(self, fs, templates, csd, pretrigger_samples, ignored_frequency_peaks=None,
                 ignore_harmonics=False):
        from . import ofnxm
        self.fs = float(fs)
        self.filt = ofnxm.NxMFilter(templates, csd, fs, pretrigger_samples, "AC",
                                    ignored_frequency_peaks, ignore_harmonics)
        self.C, self.M, self.N = self.filt.C, self.filt.M, self.filt.N
        self._templates = np.asarray(templates, dtype=np.float64)
        self.pre = int(pretrigger_samples)
        phi_fd = np.transpose(self.filt.phi, (2, 1, 0)).copy()        # [b, m, k]
        phi_fd[:, :, 0] = 0.0                                         # oftrigger.py:488
        self.phi_td = np.fft.ifft(phi_fd, axis=2).real               # oftrigger.py:489
        self.w_matrix = self.filt.P
        self.iw_matrix = self.filt.Pinv
        self.resolution = np.sqrt(np.diag(self.iw_matrix))            # oftrigger.py:496
        self.index_shift = self.pre - self.N // 2
        self.filtered = None
        self.delta_chi2 = None

    def update_trace(self, trace, padding=True):
        x = np.asarray(trace, dtype=np.float64)
        v_td = np.zeros((self.M, x.shape[-1]))
        for theta in range(self.M):                                   # oftrigger.py:656-662
            per_channel = oaconvolve(x, self.phi_td[:, theta, :], mode="same", axes=-1)
            v_td[theta] = np.sum(per_channel, axis=0)
        self.filtered = np.einsum("ij,jz->iz", self.iw_matrix / self.fs, v_td)
        self.delta_chi2 = np.einsum("iz,ij,jz->z", self.filtered, self.w_matrix, self.filtered)
        if padding:
            cut = self.N
            self.delta_chi2[:cut] = 0.0
            self.delta_chi2[-(cut) + (cut + 1) % 2:] = 0.0
        return self.filtered, self.delta_chi2

    def find_triggers_residual(self, thresh, raw, pileup_window_samples=None,
                               dynamic_function=None, positive_pulses=True, saturation=None):
        """oftrigger.py:752-845, N x M.  The filter of amplitude theta is indexed as the
        reference indexes it in this loop (``self._phi_td[theta, :]``, :800), which runs when
        the channel and amplitude counts agree."""
        raw = np.asarray(raw, dtype=np.float64)
        if saturation is None:
            saturation = [np.inf if positive_pulses else -np.inf] * self.C
        tmpl = self._templates

        def pulse_of(ti):
            amps = self.filtered[:, ti]
            trig_trace = np.zeros((self.C, self.N))
            for m in range(self.M):
                trig_trace += tmpl[:, m, :] * amps[m]
            v_td = np.zeros((self.M, self.N))
            for theta in range(self.M):
                v_td[theta] = np.sum(oaconvolve(trig_trace, self.phi_td[theta, :], mode="same",
                                                axes=-1), axis=0)
            filt = np.einsum("ij,jz->iz", self.iw_matrix / self.fs, v_td)
            return np.einsum("iz,ij,jz->z", filt, self.w_matrix, filt)

        once = lambda: self.find_triggers(thresh, pileup_window_samples=pileup_window_samples,
                                          dynamic_function=dynamic_function)
        return residual_pass(self, once, None, positive_pulses, saturation, raw, pulse_of)

    def find_triggers(self, thresh, pileup_window_msec=None, pileup_window_samples=None,
                      dynamic_function=None):
        window = 0
        if pileup_window_msec is not None:
            window = int(pileup_window_msec * self.fs / 1000)
        elif pileup_window_samples is not None:
            window = pileup_window_samples
        thr = OFTrigger.chi2_threshold(thresh, self.M)
        trig = np.where(self.delta_chi2 > thr)[0]
        idx, dchi, amp = [], [], []
        for s, e in _ranges(trig, self.delta_chi2, window, dynamic_function):
            inds = trig[s:e]
            i = inds[np.argmax(self.delta_chi2[inds])]
            idx.append(i + self.index_shift)
            dchi.append(self.delta_chi2[i])
            amp.append(self.filtered[:, i])
        idx = np.asarray(idx, dtype=np.int64)
        return {"trigger_index": idx, "trigger_time": idx / self.fs,
                "trigger_delta_chi2": np.asarray(dchi, dtype=np.float64),
                "trigger_amplitudes": np.asarray(amp, dtype=np.float64).reshape(-1, self.M),
                "chi2_threshold": thr, "pileup_window": window}
