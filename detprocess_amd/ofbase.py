"""OFBase: batch-capable stand-in for ``qetpy.OFBase`` with the method names the
reference's ``ProcessingData`` drives (``processing_data.py:278-381, 731-772``):
``set_csd, csd, add_template, phi, calc_phi, clear_signal, update_signal,
is_signal_stored, calc_signal_filt, calc_signal_filt_td``.

One-time precompute runs on the host in fp64 (``filters.build_filter``); the
per-event arithmetic runs on the GPU through ``OFPlan`` -- ``update_signal``
accepts one trace ``[N]`` or a batch ``[B, N]`` (NumPy, or a CUDA tensor for
zero-copy), and the fits return arrays ``[B]``.
"""

from math import ceil, floor

import numpy as np

from . import _lib
from .engine import OFPlan
from .filters import apply_coupling_and_notches, build_filter
from .ofnxm import NxMPlan, build_nxm_filter


def _split(channel):
    """'a|b|c' -> ['a', 'b', 'c'] (the NxM channel naming, utils.split_channel_name with '|')."""
    return [c.strip() for c in channel.split("|")]


def search_range(nb_samples, nb_pretrigger_samples, fs,
                 window_min_from_trig_usec=None, window_max_from_trig_usec=None,
                 window_min_index=None, window_max_index=None, window_policy="qetpy"):
    """Half-open rolled-bin range [lo, hi) searched by the delay fit.

    'qetpy' (default): restated QETpy >= 1.8 OFBase.get_fit_withdelay -- the
    ``*_from_trig_usec`` values win over the indices, lo = floor(pre + us fs
    1e-6), hi = ceil(pre + us fs 1e-6), [lo, hi) searched.
    'index': the features.py:1243-1344 indices, inclusive on both ends
    (SURVEY.md Appendix C-2).
    """
    N, pre = int(nb_samples), int(nb_pretrigger_samples)
    if window_policy == "qetpy":
        lo = hi = None
        if window_min_from_trig_usec is not None:
            lo = floor(pre + window_min_from_trig_usec * fs * 1e-6)
        elif window_min_index is not None:
            lo = int(window_min_index)
        if window_max_from_trig_usec is not None:
            hi = ceil(pre + window_max_from_trig_usec * fs * 1e-6)
        elif window_max_index is not None:
            hi = int(window_max_index)
    elif window_policy == "index":
        lo = None if window_min_index is None else int(window_min_index)
        hi = None if window_max_index is None else int(window_max_index) + 1
    else:
        raise ValueError('ERROR: window_policy should be "qetpy" or "index"')
    if lo is None or lo < 0:
        lo = 0
    if hi is None or hi > N:
        hi = N
    return lo, hi


class OFBase:
    def __init__(self, sample_rate, verbose=True, device=0, engine="auto", max_batch=4096):
        self._fs = float(sample_rate)
        self._verbose = verbose
        self._device = device
        self._engine = engine
        self._max_batch = max_batch
        self._nbins = None
        self._csd = {}            # channel -> effective two-sided PSD (inf at DC / notches)
        self._templates = {}      # channel -> tag -> dict(template, pre, integralnorm)
        self._tables = {}         # (channel, tag) -> FilterTables
        self._signals = {}        # channel -> traces [B, N] float32 (numpy or cuda tensor)
        self._squeeze = {}        # channel -> bool (input was a single 1-D trace)
        self._plans = {}          # (N, pre) -> OFPlan
        self._csd_nxm = {}        # 'a|b' -> dict(csd [C, C, N], coupling, peaks, harmonics)
        self._nxm_tables = {}     # ('a|b', tag) -> NxMTables
        self._nxm_plans = {}      # ('a|b', tag) -> NxMPlan
        self._fit_cache = {}

    # ------------------------------------------------------------ description
    def sample_rate(self):
        return self._fs

    def nb_samples(self):
        return self._nbins

    def fft_freqs(self):
        return None if self._nbins is None else np.fft.fftfreq(self._nbins, d=1.0 / self._fs)

    def _check_nbins(self, n, what):
        if self._nbins is None:
            self._nbins = int(n)
        elif int(n) != self._nbins:
            raise ValueError(f"ERROR: Inconsistent number of samples between {what} "
                             f"(={n}) and OF base (={self._nbins})!")

    # --------------------------------------------------------------- noise
    def set_csd(self, channel, csd, coupling="AC", ignored_frequency_peaks=None,
                ignore_harmonics=False):
        csd = np.asarray(csd)
        if csd.ndim == 3 and csd.shape[0] > 1:
            # 'a|b' channel: the N x N cross spectral density (processing_data.py:294-326)
            nchan = len(_split(channel))
            if csd.shape[0] != nchan or csd.shape[1] != nchan:
                raise ValueError(f"ERROR: csd of shape {csd.shape} for the {nchan} channel(s) "
                                 f"of {channel}")
            self._check_nbins(csd.shape[-1], "csd")
            self._csd_nxm[channel] = dict(csd=np.array(csd, dtype=np.complex128),
                                          coupling=coupling, peaks=ignored_frequency_peaks,
                                          harmonics=ignore_harmonics)
            for key in [k for k in self._nxm_tables if k[0] == channel]:
                self._nxm_tables.pop(key)
                self._drop_nxm_plan(key)
            return
        if csd.ndim == 3:
            if csd.shape[1] != 1:
                raise ValueError("ERROR: csd must be [n_channels, n_channels, samples]")
            csd = csd[0, 0]
        csd = np.real(csd).astype(np.float64)
        self._check_nbins(csd.shape[-1], "csd")
        self._csd[channel] = apply_coupling_and_notches(
            csd, self._fs, coupling, ignored_frequency_peaks, ignore_harmonics)
        for key in [k for k in self._tables if k[0] == channel]:
            self._tables.pop(key)

    def set_psd(self, channel, psd, coupling="AC", **kw):
        self.set_csd(channel, psd, coupling=coupling, **kw)

    def csd(self, channel):
        if channel in self._csd_nxm:
            return self._csd_nxm[channel]["csd"]
        return self._csd.get(channel)

    psd = csd

    # ------------------------------------------------------------ templates
    def add_template(self, channel, template, template_tag="default",
                     pretrigger_samples=None, pretrigger_msec=None, integralnorm=False,
                     overwrite=False):
        template = np.asarray(template, dtype=np.float64)
        if template.ndim == 3 and template.shape[0] == 1 and template.shape[1] == 1 \
                and "|" not in channel:
            template = template[0, 0]
        if template.ndim == 3:
            # NxM: [n_channels, m_amplitudes, samples] (oftrigger.py:375-379)
            if template.shape[0] != len(_split(channel)):
                raise ValueError(f"ERROR: template of shape {template.shape} for the "
                                 f"{len(_split(channel))} channel(s) of {channel}")
            if integralnorm:
                raise ValueError("ERROR: integralnorm is not supported for NxM templates")
        elif template.ndim != 1:
            raise ValueError("ERROR: template must be [samples] or "
                             "[n_channels, m_amplitudes, samples]")
        self._check_nbins(template.shape[-1], "template")
        tags = self._templates.setdefault(channel, {})
        if template_tag in tags and not overwrite:
            raise ValueError(f'ERROR: A template with tag "{template_tag}" already exist '
                             f'for channel {channel}! Use overwrite=True')
        if pretrigger_samples is None:
            if pretrigger_msec is not None:
                pretrigger_samples = int(round(pretrigger_msec * 1e-3 * self._fs))
            else:
                pretrigger_samples = template.shape[-1] // 2
        tags[template_tag] = dict(template=template, pre=int(pretrigger_samples),
                                  integralnorm=bool(integralnorm))
        self._tables.pop((channel, template_tag), None)
        if self._nxm_tables.pop((channel, template_tag), None) is not None:
            self._drop_nxm_plan((channel, template_tag))

    def _drop_nxm_plan(self, key):
        plan = self._nxm_plans.pop(key, None)
        if plan is not None:
            plan.close()

    def template_tags(self, channel):
        return list(self._templates.get(channel, {}).keys())

    def template(self, channel, template_tag="default"):
        t = self._templates.get(channel, {}).get(template_tag)
        return None if t is None else t["template"]

    def pretrigger_samples(self, channel, template_tag="default"):
        return self._templates[channel][template_tag]["pre"]

    def calc_phi(self, channel, template_tag=None):
        tags = [template_tag] if template_tag is not None else self.template_tags(channel)
        for tag in tags:
            if channel not in self._csd and channel not in self._csd_nxm:
                raise ValueError(f"ERROR: No csd found for channel {channel}")
            t = self._templates.get(channel, {}).get(tag)
            if t is None:
                raise ValueError(f'ERROR: No template with tag "{tag}" for channel {channel}')
            if t["template"].ndim == 3:
                c = self._csd_nxm.get(channel)
                if c is None:
                    raise ValueError(f"ERROR: No {len(_split(channel))}-channel csd found for "
                                     f"channel {channel}")
                self._nxm_tables[(channel, tag)] = build_nxm_filter(
                    t["template"], c["csd"], self._fs, t["pre"], coupling=c["coupling"],
                    ignored_frequency_peaks=c["peaks"], ignore_harmonics=c["harmonics"])
                self._drop_nxm_plan((channel, tag))
                continue
            J = self._csd[channel]
            # J already carries coupling / notches (inf) -> pass coupling="DC" so
            # build_filter does not touch bin 0 again
            self._tables[(channel, tag)] = build_filter(
                t["template"], J, self._fs, t["pre"], coupling="DC",
                integralnorm=t["integralnorm"])

    def tables(self, channel, template_tag="default"):
        if (channel, template_tag) not in self._tables:
            self.calc_phi(channel, template_tag)
        return self._tables[(channel, template_tag)]

    def phi(self, channel, template_tag="default"):
        """conj(S)/J on the one-sided grid (QETpy's phi up to its FFT normalisation);
        None until calc_phi has run (processing_data.py:379-381)."""
        if (channel, template_tag) in self._nxm_tables:
            return self._nxm_tables[(channel, template_tag)].phi
        t = self._tables.get((channel, template_tag))
        return None if t is None else t.wf * t.norm

    def iweight(self, channel, template_tag="default"):
        """Inverse of the M x M weight matrix of an NxM channel; its diagonal holds the squared
        amplitude resolutions (oftrigger.py:481, :499)."""
        return self.nxm_tables(channel, template_tag).pinv

    def nxm_tables(self, channel, template_tag="default"):
        if (channel, template_tag) not in self._nxm_tables:
            self.calc_phi(channel, template_tag)
        return self._nxm_tables[(channel, template_tag)]

    def norm(self, channel, template_tag="default"):
        return self.tables(channel, template_tag).norm

    # --------------------------------------------------------------- signal
    def clear_signal(self):
        self._signals = {}
        self._squeeze = {}
        self._fit_cache = {}

    def update_signal(self, channel, signal, calc_fft=True, **kwargs):
        """Store the trace(s) for ``channel``: [N] or [B, N]."""
        squeeze = False
        if isinstance(signal, np.ndarray):
            if signal.ndim == 1:
                signal, squeeze = signal[np.newaxis, :], True
            signal = np.ascontiguousarray(signal, dtype=np.float32)
        else:
            if signal.dim() == 1:
                signal, squeeze = signal[None, :], True
        if signal.shape[-1] != (self._nbins or signal.shape[-1]):
            raise ValueError(f"ERROR: Inconsistent number of samples between signal "
                             f"(={signal.shape[-1]}) and template/psd (={self._nbins})")
        self._check_nbins(signal.shape[-1], "signal")
        self._signals[channel] = signal
        self._squeeze[channel] = squeeze
        for key in [k for k in self._fit_cache if k[0] == channel]:
            self._fit_cache.pop(key)

    def is_signal_stored(self, channel):
        if "|" in channel and channel not in self._signals:
            return all(c in self._signals for c in _split(channel))
        return channel in self._signals

    def calc_signal_filt(self, channel, template_tag=None):
        """The FFT / filter / inverse FFT run fused with the fit on the GPU; this is
        kept so ProcessingData-style drivers (processing_data.py:769-772) work."""
        if not self.is_signal_stored(channel):
            raise ValueError(f"ERROR: no signal stored for channel {channel}")

    calc_signal_filt_td = calc_signal_filt

    # ------------------------------------------------------------------ fits
    def _plan(self, pre):
        key = (self._nbins, pre)
        if key not in self._plans:
            self._plans[key] = OFPlan(self._nbins, pre, self._fs, max_batch=self._max_batch,
                                      device=self._device, engine=self._engine)
        return self._plans[key]

    def fit(self, channel, template_tag, kind, lo=0, hi=None, outside=False,
            lowchi2_fcutoff=10000.0, interpolate=False):
        """Run one of1x1 fit on the stored batch; returns dict of float32 arrays [B]
        keyed by _lib.COL names."""
        if not self.is_signal_stored(channel):
            raise ValueError(f"ERROR: no signal stored for channel {channel}")
        ck = (channel, template_tag, kind, lo, hi, bool(outside), float(lowchi2_fcutoff),
              bool(interpolate))
        if ck in self._fit_cache:
            return self._fit_cache[ck]
        tab = self.tables(channel, template_tag)
        plan = self._plan(tab.pretrigger_samples)
        plan.reset()
        plan.set_filter(0, tab)
        sid = plan.add_search(0, kind, lo, hi, outside, lowchi2_fcutoff, interpolate)
        out = plan.process(self._signals[channel])
        if not isinstance(out, np.ndarray):
            out = out.cpu().numpy()
        off = plan.search_offset(0, sid)
        res = {name: out[:, off + j] for name, j in _lib.COL.items()}
        self._fit_cache[ck] = res
        return res

    def fit_nxm(self, channel, template_tag, lo=0, hi=None, outside=False, interpolate=False):
        """qp.OFnxm(...).calc() + get_fit_withdelay + get_fit_nodelay (algorithms.py:241-262) on
        the stored batch of an 'a|b' channel.  Returns dict: amps [B, M], t0, chi2,
        amps_nodelay [B, M], chi2_nodelay."""
        if not self.is_signal_stored(channel):
            raise ValueError(f"ERROR: no signal stored for channel {channel}")
        ck = (channel, template_tag, "nxm", lo, hi, bool(outside), bool(interpolate))
        if ck in self._fit_cache:
            return self._fit_cache[ck]
        tab = self.nxm_tables(channel, template_tag)
        key = (channel, template_tag)
        plan = self._nxm_plans.get(key)
        if plan is None:
            plan = self._nxm_plans[key] = NxMPlan(tab, max_batch=min(self._max_batch, 4096),
                                                  device=self._device)
        if channel in self._signals:                      # stored as one [B, C, N] block
            ev = self._signals[channel]
        else:
            sigs = [self._signals[c] for c in _split(channel)]
            if isinstance(sigs[0], np.ndarray):
                ev = np.stack(sigs, axis=1)
            else:
                import torch
                ev = torch.stack(sigs, dim=1)
        plan.reset_searches()
        s_nd = plan.add_search("nodelay")
        s_d = plan.add_search("delay", lo, hi, outside, interpolate)
        out = plan.process(ev)
        if not isinstance(out, np.ndarray):
            out = out.cpu().numpy()
        a0, _, c0, _ = plan.record(out, s_nd)
        a1, t1, c1, _ = plan.record(out, s_d)
        res = dict(amps=a1, t0=t1, chi2=c1, amps_nodelay=a0, chi2_nodelay=c0)
        self._fit_cache[ck] = res
        return res

    def psd_bands(self, channel, freq_ranges):
        """psd_amp: mean sqrt(folded PSD) of the stored batch per frequency range
        (algorithms.py:1001-1044); returns a list of float32 arrays [B]."""
        from .utils import get_bin_ranges
        if not self.is_signal_stored(channel):
            raise ValueError(f"ERROR: no signal stored for channel {channel}")
        sig = self._signals[channel]
        n = sig.shape[-1]
        key = ("bands", n)
        if key not in self._plans:
            self._plans[key] = OFPlan(n, 0, self._fs, max_batch=self._max_batch,
                                      device=self._device, engine="rocfft")
        plan = self._plans[key]
        plan.reset()
        ids = [plan.add_band(lo, hi) for lo, hi in get_bin_ranges(freq_ranges, n, self._fs)]
        out = plan.process(sig)
        if not isinstance(out, np.ndarray):
            out = out.cpu().numpy()
        return [out[:, plan.band_offset(i)] for i in ids]

    def squeeze(self, channel):
        if "|" in channel and channel not in self._squeeze:
            return all(self._squeeze.get(c, False) for c in _split(channel))
        return self._squeeze.get(channel, False)
