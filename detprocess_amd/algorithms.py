"""FeatureExtractors: the reference's static-method API, computed on the GPU.

Same names, arguments, returned keys, sentinel and error behaviour as
``detprocess/core/algorithms.py`` (``of1x1_nodelay`` :277-350,
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
        if interpolate:
            raise NotImplementedError("interpolate=True is not on the GPU path yet")
        r = of_base.fit(channel, template_tag, "delay", lowchi2_fcutoff=lowchi2_fcutoff)
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
        if interpolate:
            raise NotImplementedError("interpolate=True is not on the GPU path yet")
        tab = of_base.tables(channel, template_tag)
        lo, hi = search_range(tab.n_samples, tab.pretrigger_samples, of_base.sample_rate(),
                              window_min_from_trig_usec, window_max_from_trig_usec,
                              window_min_index, window_max_index, window_policy)
        r = of_base.fit(channel, template_tag, "delay", lo, hi, lgc_outside_window,
                        lowchi2_fcutoff)
        sq = of_base.squeeze(channel)
        return {f"{n}_{feature_base_name}": _maybe_scalar(r[n], sq) for n in names}

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
