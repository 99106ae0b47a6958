"""FeatureProcessing: batched counterpart of the reference's per-event loop.

The reference walks ``for event: for channel: for algorithm: extractor(...)``
(``detprocess/process/features.py:533-851``).  Here the YAML feature section is
compiled ONCE into one ``OFPlan`` per feature channel -- every of1x1 algorithm
becomes a search on a filter slot, every trace algorithm a time-domain window
-- and a whole batch of events ``[B, C, N]`` goes through the GPU in one call
per feature channel.  What is reproduced from the reference, per (channel,
algorithm): the kwargs (YAML keys minus ``run`` + ``fs``, ``nb_samples``,
``nb_pretrigger_samples``; ``features.py:762-778``), the window indices
(``:780-785``), the OF-plan key ``(nb_samples, nb_pretrigger, "<csd_tag>_<coupling>
[_<peaks>]")`` (``:794-823``), ``base_algorithm`` dispatch (``:728-730``) and the
output column names ``<feature>_<feature_channel>`` (``:842-846``).
"""

import numpy as np

from . import _lib, utils
from .config import YamlConfig
from .engine import OFPlan
from .filters import build_filter
from .ofbase import search_range

OF_ALGORITHMS = {
    # base algorithm -> (search kind, quantities emitted)   algorithms.py:344-348 etc
    "of1x1_nodelay": ("nodelay", ("amp", "chi2", "lowchi2")),
    "of1x1_unconstrained": ("delay", ("amp", "t0", "chi2", "lowchi2")),
    "of1x1_constrained": ("delay", ("amp", "t0", "chi2", "lowchi2", "chi2nopulse",
                                    "ampres", "timeres")),
}
TD_ALGORITHMS = ("baseline", "integral", "maximum", "minimum")
# Low-frequency bins (psd_amp bands, lowchi2 cut-offs) an engine keeps per trace: the FUSED kernels by
# trace length (32768: 512 in LDS + a stash up to 4096; 25000 / 20000 / 12500: 2.5 x 25 R1 bins in
# LDS), the LDS engine 1024.  A plan that needs more is compiled for the ROCFFT engine at once, instead
# of failing (engine='fused') or retrying engine after engine on every call (engine='auto').
FUSED_MAX_BIN = {32768: 4096, 25000: 1250, 20000: 1000, 12500: 625}
LDS_MAX_BIN = 1024
FUSED_MAX_BAND_BIN = FUSED_MAX_BIN[32768]      # (the 32768-sample figure, kept for callers)


def engine_bin_limit(n_samples):
    """Bins the non-ROCFFT engine that would carry an `n_samples` plan keeps for bands / lowchi2."""
    return FUSED_MAX_BIN.get(int(n_samples), LDS_MAX_BIN)

# algorithms whose base name contains one of these get an OFBase in the reference
# (processing_data.py:93-97); the ones not implemented here raise explicitly
OF_BASE_PREFIXES = ["of1x1", "of1x2x2", "of1x3x3", "ofnxm", "ofnxmx2", "psd_amp",
                    "psd_peaks", "phase"]


class _ChannelPlan:
    def __init__(self):
        self.plan = None
        self.columns = []       # (column name, row offset)
        self.energy = []        # (column name, base window off, window off, n, vb, i0, rl)
        self.external = []      # (algorithm, base, kwargs, yaml params, needs OFBase)


class FeatureProcessing:
    def __init__(self, config, filter_data, available_channels, sample_rate,
                 nb_samples=None, nb_pretrigger_samples=None, device=0, engine="auto",
                 max_batch=8192, window_policy="qetpy", external_file=None,
                 skip_unsupported=False):
        """config: YamlConfig, YAML path / text, or dict.  available_channels: the
        channel names of axis 1 of the event array, in order.  external_file: a Python
        file exposing ``class FeatureExtractors`` with user algorithms
        (features.py:248-263, 1002-1029).  skip_unsupported: channels / algorithms outside the
        hot path (of1x2x2, ofnxmx2, psd_peaks, phase) are
        skipped with a warning instead of raising, so that a full detprocess YAML such as
        examples/processing/process_example.yaml can be used as it is."""
        if isinstance(available_channels, str):
            available_channels = [available_channels]
        self._channels = list(available_channels)
        self._fs = float(sample_rate)
        if not isinstance(config, YamlConfig):
            config = YamlConfig(config, self._channels, sample_rate=sample_rate)
        self._config = config.get_config("feature")
        self._filter_data = filter_data
        self._device = device
        self._engine = engine
        self._max_batch = max_batch
        self._policy = window_policy
        self._nb_samples = nb_samples
        self._nb_pretrigger = nb_pretrigger_samples
        self._plans = None
        self._skip_unsupported = bool(skip_unsupported)
        self._ext = None
        self._ext_names = []
        if external_file is not None:
            self._ext, self._ext_names = self._load_external_extractors(external_file)

    @staticmethod
    def _load_external_extractors(external_file):
        """features.py:1002-1029 (load) and 1105-1131 (duplicates of built-ins rejected)."""
        import importlib.util
        from .algorithms import FeatureExtractors
        spec = importlib.util.spec_from_file_location("detprocess_amd._external", external_file)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        fe_ext = module.FeatureExtractors
        internal = [a for a in dir(FeatureExtractors) if a[0] != "_"]
        names = []
        for attribute in dir(fe_ext):
            if attribute[0] == "_":
                continue
            if attribute in internal:
                raise ValueError("External algorithm " + attribute
                                 + " is a duplicate from internal feature extractor!"
                                 + " This is nto allowed. You need to change name...")
            names.append(attribute)
        return fe_ext, names

    # ----------------------------------------------------------------- compile
    def _compile(self, n_samples, from_streams=False):
        """One plan per (channel, nb_samples, nb_pretrigger_samples) -- the reference's
        OF-base key (features.py:794-823; config.py:547-572 lets every algorithm carry its own
        trace length).  Float events have one length, so there every algorithm must agree
        with it; events cut from raw streams (``process_adc``) are cut per plan."""
        plans = {}
        default_n = n_samples
        weights_all = self._config.get("weights", {})
        for channel, algorithms in self._config["channels"].items():
            if not isinstance(algorithms, dict):
                continue
            feature_channel = algorithms.get("feature_channel", channel)
            if "|" in channel:
                self._compile_nxm(channel, algorithms, feature_channel, default_n, from_streams,
                                  plans)
                continue
            names, sep = utils.split_channel_name(channel, self._channels)
            idx = [self._channels.index(c) for c in names]
            w = np.ones(len(names))
            if channel in weights_all:
                for j, c in enumerate(names):
                    key = f"weight_{c}"
                    if key not in weights_all[channel]:
                        raise ValueError(f"ERROR: Missing parameter weight {key} for "
                                         f"channel {channel}!")     # processing_data.py:982-986
                    w[j] = weights_all[channel][key]
            if sep is None:
                w[:] = 1.0         # get_channel_trace ignores weights without a separator
                                   # (processing_data.py:1033-1047)
            if sep == "-":
                if len(names) != 2:
                    raise ValueError('ERROR: "-" needs exactly two channels')
                w[1] = -w[1]
            # group the running algorithms by their (nb_samples, nb_pretrigger_samples)
            groups = {}
            for algorithm, params in algorithms.items():
                if not isinstance(params, dict) or not params.get("run", False):
                    continue
                nb = params.get("nb_samples") or self._nb_samples or default_n
                npre = params.get("nb_pretrigger_samples")
                if npre is None:
                    npre = self._nb_pretrigger if self._nb_pretrigger is not None else nb // 2
                if not from_streams and nb != default_n:
                    raise ValueError(f"ERROR: Number of samples is not consistent between "
                                     f"raw data (={default_n}) and algorithm {algorithm} "
                                     f"(={nb}) for channel {channel}!")
                groups.setdefault((int(nb), int(npre)), {})[algorithm] = params
            for (n_samples, nb_pre_plan), algorithms in groups.items():
                cp = _ChannelPlan()
                slots = {}
                td_windows = {}
                pending = []
                for algorithm, params in algorithms.items():
                    base = params.get("base_algorithm", algorithm)
                    nb, npre = n_samples, nb_pre_plan
                    kwargs = {k: v for k, v in params.items() if k != "run"}
                    kwargs["fs"] = self._fs
                    kwargs.setdefault("nb_samples", nb)
                    kwargs.setdefault("nb_pretrigger_samples", npre)
                    kwargs["nb_samples"], kwargs["nb_pretrigger_samples"] = nb, npre
                    wmin, wmax = utils.get_window_indices(**kwargs)
                    if base in OF_ALGORITHMS:
                        if "template_tag" not in params:
                            raise ValueError(f'ERROR: a "template_tag" in yaml file is required '
                                             f'for channel {channel}, algorithm "{algorithm}" !')
                        csd_tag = params.get("csd_tag", "default")
                        coupling = params.get("coupling", "AC")
                        peaks = params.get("ignored_frequency_peaks")
                        if peaks is not None and not isinstance(peaks, list):
                            peaks = [peaks]
                        harm = bool(params.get("ignore_harmonics", False)) if peaks else False
                        skey = (params["template_tag"], csd_tag, coupling,
                                tuple(peaks) if peaks else None, harm,
                                bool(params.get("integralnorm", False)))
                        pending.append(("of", algorithm, base, skey, params, wmin, wmax, npre))
                    elif base in TD_ALGORITHMS:
                        pending.append(("td", algorithm, base, None, params, wmin, wmax, npre))
                    elif base == "energyabsorbed":
                        for key in ("vb", "i0", "rl"):
                            if key not in params:
                                raise ValueError(f'ERROR: energyabsorbed requires "{key}" '
                                                 f"(channel {channel})")
                        pending.append(("energy", algorithm, base, None, params, wmin, wmax, npre))
                    elif base == "psd_amp":
                        if not params.get("f_lims"):
                            raise ValueError('ERROR: "f_lims" required for algorithm psd_amps')
                        pending.append(("band", algorithm, base, None, params, wmin, wmax, npre))
                    elif base in self._ext_names:
                        # user algorithm (features.py:749-752): same injected kwargs as built-ins
                        kw = dict(kwargs)
                        kw["window_min_index"], kw["window_max_index"] = wmin, wmax
                        kw["feature_base_name"] = algorithm
                        cp.external.append((algorithm, base, kw, params,
                                            any(p in base for p in OF_BASE_PREFIXES)))
                    elif any(p in base for p in OF_BASE_PREFIXES):
                        if self._skip_unsupported:
                            import warnings
                            warnings.warn(f'algorithm "{base}" is outside the of1x1 hot path of '
                                          f"this engine: skipped")
                            continue
                        raise NotImplementedError(
                            f'algorithm "{base}" is outside the of1x1 hot path of this engine')
                    else:
                        raise ValueError(f'ERROR: Cannot find algorithm "{base}" anywhere. '
                                         f"Check feature extractor exists!")
                cp.channel, cp.feature_channel = channel, feature_channel
                cp.chan_index, cp.chan_weight, cp.chan_names = idx, w, names
                if not pending:
                    if cp.external:
                        plans[(channel, n_samples, nb_pre_plan)] = cp
                    continue
                # psd_amp bands ride on a filter slot in the FUSED engine (and only the
                # lowest bins); otherwise the general ROCFFT engine carries them
                engine = self._engine
                bands = [x for x in pending if x[0] == "band"]
                if engine != "rocfft":
                    has_of = any(x[0] == "of" for x in pending)
                    kmax = 0
                    for x in bands:
                        rng, _ = utils.cleanup_freq_ranges(x[4]["f_lims"])
                        kmax = max([kmax] + [hi for _, hi in utils.get_bin_ranges(rng, n_samples,
                                                                                  self._fs)])
                    for x in pending:                 # bins |f| <= lowchi2_fcutoff of the OF algorithms
                        if x[0] == "of":
                            fcut = float(x[4].get("lowchi2_fcutoff", 10000))
                            kmax = max(kmax, int(np.floor(fcut * n_samples / self._fs)) + 1)
                    if (bands and not has_of) or kmax > engine_bin_limit(n_samples):
                        engine = "rocfft"
                plan = OFPlan(n_samples, nb_pre_plan, self._fs, max_batch=self._max_batch,
                              device=self._device, engine=engine)
                if len(self._channels) > 1 or len(idx) > 1 or w[0] != 1.0:
                    plan.set_channels(len(self._channels), idx, w)
                cols = []
                for kind, algorithm, base, skey, params, wmin, wmax, npre in pending:
                    if kind == "of":
                        if skey not in slots:
                            slot = len(slots)
                            tag, csd_tag, coupling, peaks, harm, inorm = skey
                            # the filter file is looked up with the YAML channel expression
                            # itself ('A+B' needs its own template / csd entry) and raises
                            # 'Channel ... not available' otherwise: processing_data.py:295, 345
                            fchan = channel
                            template, _, tmeta = self._filter_data.get_template(
                                fchan, tag=tag, return_metadata=True)
                            csd, _, cmeta = self._filter_data.get_csd(
                                fchan, tag=csd_tag, fold=False, return_metadata=True)
                            if "sample_rate" in cmeta and cmeta["sample_rate"] != self._fs:
                                raise ValueError(f"Sample rate is not consistent between raw "
                                                 f"data and csd for channel {channel}!")
                            if n_samples != csd.shape[-1]:
                                raise ValueError(
                                    f"Number of samples is not consistent between raw data "
                                    f"(={n_samples}) and csd (={csd.shape[-1]})for channel "
                                    f"{channel}, algorithm {algorithm}!")
                            if n_samples != template.shape[-1]:
                                raise ValueError(
                                    f'Number of samples is not consistent between raw data and '
                                    f'template ("{tag}") for channel {channel}, algorithm '
                                    f"{algorithm}!")
                            pre_t = int(tmeta.get("nb_pretrigger_samples", npre))
                            if pre_t != nb_pre_plan:
                                raise ValueError("ERROR: template pretrigger differs from the "
                                                 "trace pretrigger")
                            tables = build_filter(template, csd, self._fs, pre_t, coupling,
                                                  list(peaks) if peaks else None, harm, inorm)
                            plan.set_filter(slot, tables)
                            slots[skey] = slot
                        slot = slots[skey]
                        skind, qtys = OF_ALGORITHMS[base]
                        fcut = float(params.get("lowchi2_fcutoff", 10000))
                        interp = bool(params.get("interpolate", False)) and skind == "delay"
                        if base == "of1x1_constrained":
                            lo, hi = search_range(n_samples, nb_pre_plan, self._fs,
                                                  params.get("window_min_from_trig_usec"),
                                                  params.get("window_max_from_trig_usec"),
                                                  wmin, wmax, self._policy)
                            sid = plan.add_search(slot, "delay", lo, hi,
                                                  bool(params.get("lgc_outside_window", False)),
                                                  fcut, interp)
                        else:
                            sid = plan.add_search(slot, skind, lowchi2_fcutoff=fcut,
                                                  interpolate=interp)
                        cols.append(("of", slot, sid, qtys, algorithm))
                    elif kind == "band":
                        rng, rnames = utils.cleanup_freq_ranges(params["f_lims"])
                        for (klo, khi), rname in zip(utils.get_bin_ranges(rng, n_samples, self._fs),
                                                     rnames):
                            cols.append(("band", plan.add_band(klo, khi), f"{algorithm}_{rname}"))
                    elif kind == "energy":
                        if wmin < 1:
                            raise ValueError("ERROR: energyabsorbed needs a window starting after "
                                             "sample 0 (its baseline is mean(trace[:window_min]))")
                        for key in ((0, wmin), (wmin, wmax)):
                            if key not in td_windows:
                                td_windows[key] = plan.add_tdwindow(*key)
                        cols.append(("energy", td_windows[(0, wmin)], td_windows[(wmin, wmax)],
                                     wmax - wmin, params, algorithm))
                    else:
                        hi = wmax              # end-exclusive slice trace[wmin:wmax]
                        key = (wmin, hi)
                        if key not in td_windows:
                            td_windows[key] = plan.add_tdwindow(wmin, hi)
                        cols.append(("td", td_windows[key], base, algorithm))
                for c in cols:
                    if c[0] == "of":
                        _, slot, sid, qtys, algorithm = c
                        off = plan.search_offset(slot, sid)
                        for q in qtys:
                            cp.columns.append((f"{q}_{algorithm}_{feature_channel}",
                                               off + _lib.COL[q]))
                    elif c[0] == "band":
                        cp.columns.append((f"{c[2]}_{feature_channel}", plan.band_offset(c[1])))
                    elif c[0] == "energy":
                        _, wb, ww, nwin, params, algorithm = c
                        cp.energy.append((f"{algorithm}_{feature_channel}", plan.tdwindow_offset(wb),
                                          plan.tdwindow_offset(ww), nwin, float(params["vb"]),
                                          float(params["i0"]), float(params["rl"])))
                    else:
                        _, wid, base, algorithm = c
                        off = plan.tdwindow_offset(wid)
                        cp.columns.append((f"{algorithm}_{feature_channel}", off + _lib.TD[base]))
                cp.plan = plan
                plans[(channel, n_samples, nb_pre_plan)] = cp
        self._plans = plans
        self._compiled_n = default_n
        self._compiled_streams = bool(from_streams)

    def _unsupported(self, what):
        if self._skip_unsupported:
            import warnings
            warnings.warn(f"{what} is outside the hot path of this engine: skipped")
            return
        raise NotImplementedError(f"{what} is outside the hot path of this engine")

    def _compile_nxm(self, channel, algorithms, feature_channel, default_n, from_streams, plans):
        """An ``a|b`` channel (features.py:692-716 passes the name through; the extractor splits
        it, algorithms.py:188-192): every running ``ofnxm`` algorithm becomes a delay search of
        an NxM plan; algorithms with the same trace length, template and csd tags share the
        plan, its transforms and its no-delay fit."""
        from .ofnxm import NxMPlan, build_nxm_filter
        names = [c.strip() for c in channel.split("|")]
        for c in names:
            if c not in self._channels:
                raise ValueError(f'ERROR: Channel "{c}" of "{channel}" is not available! '
                                 f"Available channels: {self._channels}")
        idx = [self._channels.index(c) for c in names]
        groups = {}
        for algorithm, params in algorithms.items():
            if not isinstance(params, dict) or not params.get("run", False):
                continue
            base = params.get("base_algorithm", algorithm)
            if base != "ofnxm":
                self._unsupported(f'algorithm "{base}" of the multi-channel entry "{channel}"')
                continue
            nb = params.get("nb_samples") or self._nb_samples or default_n
            npre = params.get("nb_pretrigger_samples")
            if npre is None:
                npre = self._nb_pretrigger if self._nb_pretrigger is not None else nb // 2
            if not from_streams and nb != default_n:
                raise ValueError(f"ERROR: Number of samples is not consistent between raw data "
                                 f"(={default_n}) and algorithm {algorithm} (={nb}) for channel "
                                 f"{channel}!")
            if "template_tag" not in params:
                raise ValueError(f'ERROR: a "template_tag" in yaml file is required for channel '
                                 f'{channel}, algorithm "{algorithm}" !')
            peaks = params.get("ignored_frequency_peaks")
            if peaks is not None and not isinstance(peaks, list):
                peaks = [peaks]
            harm = bool(params.get("ignore_harmonics", False)) if peaks else False
            skey = (int(nb), int(npre), params["template_tag"], params.get("csd_tag", "default"),
                    params.get("coupling", "AC"), tuple(peaks) if peaks else None, harm)
            groups.setdefault(skey, []).append((algorithm, params))
        for skey, algos in groups.items():
            nb, npre, tag, csd_tag, coupling, peaks, harm = skey
            template, _, tmeta = self._filter_data.get_template(channel, tag=tag,
                                                                return_metadata=True)
            csd, _, cmeta = self._filter_data.get_csd(channel, tag=csd_tag, fold=False,
                                                      return_metadata=True)
            if "sample_rate" in cmeta and cmeta["sample_rate"] != self._fs:
                raise ValueError(f"Sample rate is not consistent between raw data and csd for "
                                 f"channel {channel}!")
            if nb != csd.shape[-1]:
                raise ValueError(f"Number of samples is not consistent between raw data (={nb}) "
                                 f"and csd (={csd.shape[-1]})for channel {channel}!")
            if nb != template.shape[-1]:
                raise ValueError(f'Number of samples is not consistent between raw data and '
                                 f'template ("{tag}") for channel {channel}!')
            pre_t = int(tmeta.get("nb_pretrigger_samples", npre))
            if pre_t != npre:
                raise ValueError("ERROR: template pretrigger differs from the trace pretrigger")
            tables = build_nxm_filter(template, csd, self._fs, pre_t, coupling,
                                      list(peaks) if peaks else None, harm)
            plan = NxMPlan(tables, max_batch=min(self._max_batch, 4096), device=self._device)
            plan.set_channels(len(self._channels), idx)
            cp = _ChannelPlan()
            cp.nxm = True
            cp.channel, cp.feature_channel = channel, feature_channel
            cp.chan_index, cp.chan_names = idx, names
            m = tables.n_tmpl
            s_nd = plan.add_search("nodelay")
            for algorithm, params in algos:
                amp_names = params.get("amplitude_names")
                if amp_names is None:
                    amp_names = [f"amp{i + 1}" for i in range(m)]
                elif isinstance(amp_names, str):
                    amp_names = [amp_names]
                if len(amp_names) != m:
                    raise ValueError(f'ERROR: Wrong length for "amplitude_names" argument. '
                                     f"Expecting {m} name for  channel {channel}, algorithm "
                                     f'"{algorithm}"')                 # algorithms.py:221-226
                kwargs = {k: v for k, v in params.items() if k != "run"}
                kwargs["fs"] = self._fs
                kwargs["nb_samples"], kwargs["nb_pretrigger_samples"] = nb, npre
                wmin, wmax = utils.get_window_indices(**kwargs)
                lo, hi = search_range(nb, npre, self._fs, params.get("window_min_from_trig_usec"),
                                      params.get("window_max_from_trig_usec"), wmin, wmax,
                                      self._policy)
                s_d = plan.add_search("delay", lo, hi,
                                      bool(params.get("lgc_outside_window", False)),
                                      bool(params.get("interpolate_t0", False)))
                od, on = s_d * (m + 3), s_nd * (m + 3)
                cp.columns.append((f"chi2_{algorithm}_constrained_{feature_channel}", od + m + 1))
                cp.columns.append((f"t0_{algorithm}_constrained_{feature_channel}", od + m))
                for i, an in enumerate(amp_names):
                    cp.columns.append((f"{an}_{algorithm}_constrained_{feature_channel}", od + i))
                cp.columns.append((f"chi2_{algorithm}_nodelay_{feature_channel}", on + m + 1))
                for i, an in enumerate(amp_names):
                    cp.columns.append((f"{an}_{algorithm}_nodelay_{feature_channel}", on + i))
            cp.plan = plan
            plans[(channel,) + skey] = cp

    # ----------------------------------------------------------------- process
    def columns(self):
        return [name for cp in (self._plans or {}).values()
                for name in [c[0] for c in cp.columns] + [e[0] for e in cp.energy]]

    def process(self, traces, valid=None, as_dataframe=True):
        """traces: float32 [B, C, N] (C = len(available_channels)) or [B, N] when there
        is one channel; NumPy or CUDA tensor.  valid: optional [B] mask (0 -> every
        feature of the event is -999999.0).  Returns a pandas DataFrame (or dict)."""
        shape = tuple(traces.shape)
        if len(shape) == 2:
            if len(self._channels) != 1:
                raise ValueError("ERROR: traces must be [B, C, N]")
            traces = traces.reshape(shape[0], 1, shape[1])
            shape = tuple(traces.shape)
        if shape[1] != len(self._channels):
            raise ValueError(f"ERROR: traces have {shape[1]} channels, expected "
                             f"{len(self._channels)}")
        if (self._plans is None or self._compiled_n != shape[2]
                or getattr(self, "_compiled_streams", False)):
            self._compile(shape[2])
        def run(cp):
            tr = traces
            if getattr(cp, "nxm", False):
                return cp.plan.process(tr, valid=valid)
            if len(self._channels) == 1 and cp.plan.n_channels == 1:
                tr = traces.reshape(shape[0], shape[2])
            return cp.plan.process(tr, valid=valid)
        return self._collect(run, as_dataframe, traces=traces, valid=valid)

    def plans(self, n_samples=None):
        """{(channel, nb_samples, nb_pretrigger_samples): plan} of the compiled configuration
        (compiles for ``n_samples`` first when given)."""
        if n_samples is not None and (self._plans is None or self._compiled_n != n_samples
                                      or getattr(self, "_compiled_streams", False)):
            self._compile(int(n_samples))
        return {k: cp.plan for k, cp in (self._plans or {}).items() if cp.plan is not None}

    def device_columns(self):
        """{plan key: [(column name, float offset in the plan's output row)]} -- the map from the
        device-resident rows of ``process_device`` to the reference's column names."""
        return {k: list(cp.columns) for k, cp in (self._plans or {}).items() if cp.plan is not None}

    def process_device(self, traces, valid=None, outs=None):
        """``process`` without the trip to the host: traces is a CUDA tensor [B, C, N]; every
        channel plan is launched once on it and its float32 [B, row] output stays on the
        device.  Returns {plan key: tensor}; ``outs`` (same keys) supplies preallocated outputs.
        External (host-side) extractors and energyabsorbed's host finish are not run here."""
        shape = tuple(traces.shape)
        if len(shape) != 3 or shape[1] != len(self._channels):
            raise ValueError(f"ERROR: traces must be [B, {len(self._channels)}, N]")
        if (self._plans is None or self._compiled_n != shape[2]
                or getattr(self, "_compiled_streams", False)):
            self._compile(shape[2])
        res = {}
        for key, cp in self._plans.items():
            if cp.plan is None:
                continue
            tr = traces
            if (not getattr(cp, "nxm", False) and len(self._channels) == 1
                    and cp.plan.n_channels == 1):
                tr = traces.reshape(shape[0], shape[2])
            if getattr(cp, "nxm", False):
                res[key] = cp.plan.process(tr, valid=valid)
            else:
                res[key] = cp.plan.process(tr, valid=valid,
                                           out=None if outs is None else outs.get(key))
        return res

    def process_adc(self, adc, trigger_index, scale, offset, n_samples=None, as_dataframe=True):
        """Events cut on the GPU out of continuous raw-data streams (SURVEY.md 8f rank 2;
        processing_data.py:640-656, 674-684).  adc: int16 [C, n_stream] (C =
        len(available_channels)), NumPy or CUDA tensor; trigger_index: int64 [B];
        scale / offset: per channel ADC -> amps (pytesio ``adctoamp``); n_samples: trace
        length of algorithms that do not carry their own (default: the configured
        ``nb_samples``); algorithms with their own ``nb_samples`` / ``nb_pretrigger_samples``
        get their own window around the same trigger, as in the reference
        (processing_data.py:640-656).  Windows that do not fit in the stream come back as
        -999999.0."""
        n = int(n_samples or self._nb_samples or 0)
        if n <= 0:
            raise ValueError("ERROR: process_adc needs the trace length (n_samples= or "
                             "nb_samples in the configuration)")
        if (self._plans is None or self._compiled_n != n
                or not getattr(self, "_compiled_streams", False)):
            self._compile(n, from_streams=True)
        if adc.shape[0] != len(self._channels):
            raise ValueError(f"ERROR: adc has {adc.shape[0]} channels, expected "
                             f"{len(self._channels)}")
        sc = np.broadcast_to(np.asarray(scale, dtype=np.float64), (len(self._channels),))
        of = np.broadcast_to(np.asarray(offset, dtype=np.float64), (len(self._channels),))

        def run(cp):
            if getattr(cp, "nxm", False):
                return cp.plan.process_adc(adc, trigger_index, sc, of)
            if cp.plan.n_channels == 1 and len(self._channels) == 1:
                return cp.plan.process_adc(adc, trigger_index, sc[:1], of[:1])
            return cp.plan.process_adc(adc, trigger_index, sc, of)
        return self._collect(run, as_dataframe)

    def _run_external(self, cp, traces, valid, result):
        """User algorithms of an ``external_file`` (features.py:749-752, 826-839).  Trace
        algorithms are called once per event with the float64 channel trace, as the
        reference does; algorithms whose base name asks for an OF base
        (processing_data.py:93-97) get a ``detprocess_amd.OFBase`` holding the whole batch."""
        if traces is None:
            raise NotImplementedError("external extractors need the float events "
                                      "(FeatureProcessing.process)")
        tr = traces.cpu().numpy() if not isinstance(traces, np.ndarray) else traces
        B = tr.shape[0]
        comb = np.zeros((B, tr.shape[2]), dtype=np.float64)
        for j, wj in zip(cp.chan_index, cp.chan_weight):
            comb += float(wj) * tr[:, j, :].astype(np.float64)
        ok = np.ones(B, dtype=bool) if valid is None else np.asarray(valid).astype(bool)
        for algorithm, base, kw, params, needs_of in cp.external:
            extractor = getattr(self._ext, base)
            if needs_of:
                from .ofbase import OFBase
                fchan = cp.channel        # exact expression, as processing_data.py:295, 345
                ob = OFBase(self._fs, device=self._device, engine=self._engine)
                csd, _, _ = self._filter_data.get_csd(fchan, tag=params.get("csd_tag", "default"),
                                                      fold=False, return_metadata=True)
                ob.set_csd(cp.channel, csd, coupling=params.get("coupling", "AC"),
                           ignored_frequency_peaks=params.get("ignored_frequency_peaks"),
                           ignore_harmonics=bool(params.get("ignore_harmonics", False)))
                if "template_tag" in params:
                    template, _, _ = self._filter_data.get_template(
                        fchan, tag=params["template_tag"], return_metadata=True)
                    ob.add_template(cp.channel, template, template_tag=params["template_tag"],
                                    pretrigger_samples=kw["nb_pretrigger_samples"])
                    ob.calc_phi(cp.channel, params["template_tag"])
                ob.update_signal(cp.channel, comb.astype(np.float32), calc_fft=True)
                feats = extractor(cp.channel, ob, **kw)
                for name, val in feats.items():
                    arr = np.broadcast_to(np.asarray(val, dtype=np.float64), (B,)).copy()
                    arr[~ok] = -999999.0
                    result[f"{name}_{cp.feature_channel}"] = arr
                continue
            cols = {}
            for b in range(B):
                if not ok[b]:
                    continue
                feats = extractor(comb[b], **kw)
                for name, val in feats.items():
                    cols.setdefault(name, np.full(B, -999999.0))[b] = val
            for name, arr in cols.items():
                result[f"{name}_{cp.feature_channel}"] = arr

    def _collect(self, run, as_dataframe, traces=None, valid=None):
        result = {}
        for channel, cp in self._plans.items():
            if cp.external:
                self._run_external(cp, traces, valid, result)
            if cp.plan is None:
                continue
            out = run(cp)
            if not isinstance(out, np.ndarray):
                out = out.cpu().numpy()
            for name, off in cp.columns:
                result[name] = out[:, off].astype(np.float64)
            for name, ob, ow, nwin, vb, i0, rl in cp.energy:
                from .algorithms import energy_absorbed_value
                o64 = out.astype(np.float64)
                val = energy_absorbed_value(o64[:, ob:ob + 8], o64[:, ow:ow + 8], nwin,
                                            self._fs, vb, i0, rl)
                bad = out[:, ow] == -999999.0
                result[name] = np.where(bad, -999999.0, val)
        if as_dataframe:
            import pandas as pd
            return pd.DataFrame(result)
        return result
