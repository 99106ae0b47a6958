"""FeatureExtractors: the reference's static-method API, computed on the GPU.

Same names, arguments, returned keys, sentinel and error behaviour as
``detprocess/core/algorithms.py`` (``ofnxm`` :141-274, ``of1x1_nodelay`` :277-350,
``of1x1_unconstrained`` :354-432, ``of1x1_constrained`` :435-570, ``baseline``
:650-704, ``integral`` :708-765, ``maximum`` :770-824, ``minimum`` :829-885).
Differences: ``of_base`` is ``detprocess_amd.OFBase`` and traces may be batches
(``[B, N]``): scalars come back for a single trace, arrays ``[B]`` for a batch.
"""

import numpy as np

from .engine import OFPlan
from .ofbase import search_range

SENTINEL = -999999.0


def _maybe_scalar(arr, squeeze):
    return float(arr[0]) if squeeze else np.asarray(arr, dtype=np.float64)


_TD_PLANS = {}


def _td_record(trace, fs, windows):
    """All time-domain records ([B, 8] float64 per window) for a list of slices."""
    squeeze = False
    if isinstance(trace, np.ndarray):
        if trace.ndim == 1:
            trace, squeeze = trace[np.newaxis, :], True
        trace = np.ascontiguousarray(trace, dtype=np.float32)
        device = 0
    else:
        if trace.dim() == 1:
            trace, squeeze = trace[None, :], True
        device = trace.device.index if trace.is_cuda else 0
    n = trace.shape[-1]
    key = (n, float(fs), device)
    plan = _TD_PLANS.get(key)
    if plan is None:
        plan = _TD_PLANS[key] = OFPlan(n, 0, fs, max_batch=4096, device=device)
    plan.reset()
    wids = [plan.add_tdwindow(lo, hi) for lo, hi in windows]
    out = plan.process(trace)
    if not isinstance(out, np.ndarray):
        out = out.cpu().numpy()
    recs = [out[:, plan.tdwindow_offset(w): plan.tdwindow_offset(w) + 8].astype(np.float64)
            for w in wids]
    return recs, squeeze


def energy_absorbed_value(rec_base, rec_win, n, fs, vb, i0, rl):
    """energyabsorbed (algorithms.py:938-943) from two GPU window records
    (baseline window trace[:lo], integration window trace[lo:hi], n = hi - lo):
    i = x - b; p0 = i (vb - 2 i0 rl) - i^2 rl; trapz(p0)/fs with
    sum(i) = S - n b and sum(i^2) = Q - 2 b S + n b^2."""
    b = rec_base[:, 0]
    S, Q, first, last = rec_win[:, 4], rec_win[:, 5], rec_win[:, 6], rec_win[:, 7]
    c1 = vb - 2.0 * i0 * rl
    si = S - n * b
    si2 = Q - 2.0 * b * S + n * b * b
    p0 = lambda x: (x - b) * c1 - (x - b) ** 2 * rl
    return (c1 * si - rl * si2 - 0.5 * (p0(first) + p0(last))) / fs


def _td_feature(trace, which, fs, window_min_index, window_max_index):
    """baseline / integral / maximum / minimum of trace[lo:hi] on the GPU."""
    squeeze = False
    if isinstance(trace, np.ndarray):
        if trace.ndim == 1:
            trace, squeeze = trace[np.newaxis, :], True
        trace = np.ascontiguousarray(trace, dtype=np.float32)
        device = 0
    else:
        if trace.dim() == 1:
            trace, squeeze = trace[None, :], True
        device = trace.device.index if trace.is_cuda else 0
    n = trace.shape[-1]
    lo = 0 if window_min_index is None else int(window_min_index)
    hi = n - 1 if window_max_index is None else int(window_max_index)   # :694-695
    key = (n, float(fs), device)
    plan = _TD_PLANS.get(key)
    if plan is None:
        plan = _TD_PLANS[key] = OFPlan(n, 0, fs, max_batch=4096, device=device)
    plan.reset()
    wid = plan.add_tdwindow(lo, hi)
    out = plan.process(trace)
    if not isinstance(out, np.ndarray):
        out = out.cpu().numpy()
    col = {"baseline": 0, "integral": 1, "maximum": 2, "minimum": 3}[which]
    return _maybe_scalar(out[:, plan.tdwindow_offset(wid) + col], squeeze)


class FeatureExtractors:
    """Static methods, one per feature algorithm, returning {feature_name: value}."""

    @staticmethod
    def ofnxm(channel, of_base, available_channels=None, feature_base_name="ofnxm",
              template_tag=None, amplitude_names=None, window_min_from_trig_usec=None,
              window_max_from_trig_usec=None, window_min_index=None, window_max_index=None,
              lgc_outside_window=False, lowchi2_fcutoff=10000, interpolate_t0=False, **kwargs):
        """NxM optimal filter of an ``a|b`` channel (algorithms.py:141-274): the windowed delay
        fit (``*_constrained``) and the no-delay fit (``*_nodelay``) of M amplitudes."""
        if template_tag is None:
            raise ValueError(f'ERROR: Missing "template_tag" argument for channel {channel}, '
                             f'algorithm "{feature_base_name}"')
        template = of_base.template(channel, template_tag=template_tag)
        if template is None:
            raise ValueError(f'ERROR: Missing template for channel {channel}, tag '
                             f'"{template_tag}", algorithm "{feature_base_name}"')
        if template.ndim != 3:
            raise ValueError(f"ERROR: the template of channel {channel} must be "
                             f"[n_channels, n_amplitudes, samples]")
        ntmps = template.shape[1]
        if amplitude_names is None:
            amplitude_names = [f"amp{i + 1}" for i in range(ntmps)]
        else:
            if isinstance(amplitude_names, str):
                amplitude_names = [amplitude_names]
            if len(amplitude_names) != ntmps:
                raise ValueError(f'ERROR: Wrong length for "amplitude_names" argument. Expecting '
                                 f'{ntmps} name for  channel {channel}, algorithm '
                                 f'"{feature_base_name}"')
        ret = {f"chi2_{feature_base_name}_constrained": SENTINEL,
               f"t0_{feature_base_name}_constrained": SENTINEL}
        for name in amplitude_names:
            ret[f"{name}_{feature_base_name}_constrained"] = SENTINEL
        ret[f"chi2_{feature_base_name}_nodelay"] = SENTINEL
        for name in amplitude_names:
            ret[f"{name}_{feature_base_name}_nodelay"] = SENTINEL
        if not of_base.is_signal_stored(channel):
            return ret
        lo, hi = search_range(template.shape[-1], of_base.pretrigger_samples(channel, template_tag),
                              of_base.sample_rate(), window_min_from_trig_usec,
                              window_max_from_trig_usec, window_min_index, window_max_index)
        r = of_base.fit_nxm(channel, template_tag, lo, hi, bool(lgc_outside_window),
                            interpolate=bool(interpolate_t0))
        sq = of_base.squeeze(channel)
        ret[f"chi2_{feature_base_name}_constrained"] = _maybe_scalar(r["chi2"], sq)
        ret[f"t0_{feature_base_name}_constrained"] = _maybe_scalar(r["t0"], sq)
        for i, name in enumerate(amplitude_names):
            ret[f"{name}_{feature_base_name}_constrained"] = _maybe_scalar(r["amps"][:, i], sq)
        ret[f"chi2_{feature_base_name}_nodelay"] = _maybe_scalar(r["chi2_nodelay"], sq)
        for i, name in enumerate(amplitude_names):
            ret[f"{name}_{feature_base_name}_nodelay"] = _maybe_scalar(r["amps_nodelay"][:, i], sq)
        return ret

    @staticmethod
    def of1x1_nodelay(channel, of_base, template_tag=None, lowchi2_fcutoff=10000,
                      feature_base_name="of1x1_nodelay", **kwargs):
        if template_tag is None:
            raise ValueError("ERROR: Template tag required for OF 1x1")
        names = ("amp", "chi2", "lowchi2")
        if not of_base.is_signal_stored(channel):
            return {f"{n}_{feature_base_name}": SENTINEL for n in names}
        r = of_base.fit(channel, template_tag, "nodelay", lowchi2_fcutoff=lowchi2_fcutoff)
        sq = of_base.squeeze(channel)
        return {f"{n}_{feature_base_name}": _maybe_scalar(r[n], sq) for n in names}

    @staticmethod
    def of1x1_unconstrained(channel, of_base, template_tag="default", interpolate=False,
                            lowchi2_fcutoff=10000,
                            feature_base_name="of1x1_unconstrained", **kwargs):
        names = ("amp", "t0", "chi2", "lowchi2")
        if not of_base.is_signal_stored(channel):
            return {f"{n}_{feature_base_name}": SENTINEL for n in names}
        r = of_base.fit(channel, template_tag, "delay", lowchi2_fcutoff=lowchi2_fcutoff,
                        interpolate=bool(interpolate))
        sq = of_base.squeeze(channel)
        return {f"{n}_{feature_base_name}": _maybe_scalar(r[n], sq) for n in names}

    @staticmethod
    def of1x1_constrained(channel, of_base, template_tag="default",
                          window_min_from_trig_usec=None, window_max_from_trig_usec=None,
                          window_min_index=None, window_max_index=None,
                          lgc_outside_window=False, interpolate=False,
                          lowchi2_fcutoff=10000, feature_base_name="of1x1_constrained",
                          window_policy="qetpy", **kwargs):
        names = ("amp", "t0", "chi2", "lowchi2", "chi2nopulse", "ampres", "timeres")
        if not of_base.is_signal_stored(channel):
            return {f"{n}_{feature_base_name}": SENTINEL for n in names}
        tab = of_base.tables(channel, template_tag)
        lo, hi = search_range(tab.n_samples, tab.pretrigger_samples, of_base.sample_rate(),
                              window_min_from_trig_usec, window_max_from_trig_usec,
                              window_min_index, window_max_index, window_policy)
        r = of_base.fit(channel, template_tag, "delay", lo, hi, lgc_outside_window,
                        lowchi2_fcutoff, bool(interpolate))
        sq = of_base.squeeze(channel)
        return {f"{n}_{feature_base_name}": _maybe_scalar(r[n], sq) for n in names}

    @staticmethod
    def energyabsorbed(trace, fs, vb, i0, rl, window_min_index=None, window_max_index=None,
                       feature_base_name="energyabsorbed", **kwargs):
        """algorithms.py:889-949: baseline = mean(trace[:lo]); integral of
        p0 = i (vb - 2 i0 rl) - i^2 rl over trace[lo:hi] (trapezoid, dx = 1/fs)."""
        if trace is None or (hasattr(trace, "size") and np.size(trace) == 0):
            return {feature_base_name: SENTINEL}
        lo, hi = int(window_min_index), int(window_max_index)
        if lo < 1:
            raise ValueError("ERROR: energyabsorbed needs window_min_index >= 1 "
                             "(the baseline is the mean of trace[:window_min_index])")
        (rb, rw), squeeze = _td_record(trace, fs, [(0, lo), (lo, hi)])
        return {feature_base_name: _maybe_scalar(
            energy_absorbed_value(rb, rw, hi - lo, fs, vb, i0, rl), squeeze)}

    @staticmethod
    def psd_amp(channel, of_base, f_lims=[], feature_base_name="psd_amp", **kwargs):
        """algorithms.py:952-1044: average sqrt(folded PSD) of the stored signal in
        each [f_low, f_high]; keys '<base>_<low>_<high>'."""
        if not f_lims:
            raise ValueError('ERROR: "f_lims" required for algorithm psd_amps')
        from .utils import cleanup_freq_ranges
        ranges, names = cleanup_freq_ranges(f_lims)
        if not of_base.is_signal_stored(channel):
            return {f"{feature_base_name}_{n}": SENTINEL for n in names}
        vals = of_base.psd_bands(channel, ranges)
        sq = of_base.squeeze(channel)
        return {f"{feature_base_name}_{n}": _maybe_scalar(v, sq) for n, v in zip(names, vals)}

    @staticmethod
    def baseline(trace, window_min_index=None, window_max_index=None,
                 feature_base_name="baseline", **kwargs):
        if trace is None or (hasattr(trace, "size") and np.size(trace) == 0):
            return {feature_base_name: SENTINEL}
        return {feature_base_name: _td_feature(trace, "baseline", kwargs.get("fs", 1.0),
                                               window_min_index, window_max_index)}

    @staticmethod
    def integral(trace, fs, window_min_index=None, window_max_index=None,
                 feature_base_name="integral", **kwargs):
        if trace is None or (hasattr(trace, "size") and np.size(trace) == 0):
            return {feature_base_name: SENTINEL}
        return {feature_base_name: _td_feature(trace, "integral", fs, window_min_index,
                                               window_max_index)}

    @staticmethod
    def maximum(trace, window_min_index=None, window_max_index=None,
                feature_base_name="maximum", **kwargs):
        if trace is None or (hasattr(trace, "size") and np.size(trace) == 0):
            return {feature_base_name: SENTINEL}
        return {feature_base_name: _td_feature(trace, "maximum", kwargs.get("fs", 1.0),
                                               window_min_index, window_max_index)}

    @staticmethod
    def minimum(trace, window_min_index=None, window_max_index=None,
                feature_base_name="minimum", **kwargs):
        if trace is None or (hasattr(trace, "size") and np.size(trace) == 0):
            return {feature_base_name: SENTINEL}
        return {feature_base_name: _td_feature(trace, "minimum", kwargs.get("fs", 1.0),
                                               window_min_index, window_max_index)}
