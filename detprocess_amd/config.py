"""YamlConfig: the reference's YAML surface for feature processing.

Mirrors ``detprocess/process/config.py`` (``_read_config`` :117-277,
``_configure_features`` :409-644, duplicate-key-rejecting loader :666-684,
obsolete-key map :71-79): same keys in, same dictionary out
(``get_config('feature')`` -> ``overall / channels / channel_list /
traces_config / weights``).  Trigger / salting / didv / noise / template
sections are carried through unprocessed (out of scope).
"""

import copy

import yaml
from yaml.loader import SafeLoader

from . import utils

CONFIGURATION_FIELDS = ["salting", "feature", "didv", "noise", "template", "trigger"]
GLOBAL_PARAMETERS = ["filter_file", "didv_file"]
FEATURE_OVERALL = ["trace_length_samples", "pretrigger_length_samples",
                   "trace_length_msec", "pretrigger_length_msec"]
OBSOLETE_KEYS = {          # config.py:71-79
    "trigger_name": "trigger_channel",
    "nb_samples": "trace_length_samples",
    "nb_pretrigger_samples": "pretrigger_length_samples",
    "template_time_tags": "template_group_ids",
    "psd_tag": "csd_tag",
    "noise_tag": "csd_tag",
    "deadtime_salt": "do_salt_deadtime",
}


class _UniqueKeyLoader(SafeLoader):
    """PyYAML silently keeps the last duplicate key; the reference refuses
    (config.py:666-684).  Duplicates are looked for on the key nodes as written, BEFORE the stock
    constructor expands merge keys ('<<: *base' followed by an overriding key is legal YAML and
    not a duplicate); the mapping itself is then built by the stock constructor."""

    def construct_mapping(self, node, deep=False):
        seen = set()
        for key_node, _ in node.value:
            if key_node.tag == "tag:yaml.org,2002:merge":
                continue
            key = self.construct_object(key_node, deep=True)
            try:
                dup = key in seen
            except TypeError:                  # unhashable key: left to the stock constructor
                continue
            if dup:
                raise ValueError(f'ERROR: Duplicate key "{key}" found in the yaml file for '
                                 f'same channel and algorithm. This is not allowed to '
                                 f'avoid unwanted configuration!')
            seen.add(key)
        return super().construct_mapping(node, deep=deep)


def _rename_keys(d, old, new):
    if not isinstance(d, dict):
        return d
    for key in list(d.keys()):
        if isinstance(d[key], dict):
            _rename_keys(d[key], old, new)
        if key == old:
            d[new] = d.pop(old)
    return d


def _load(source):
    if isinstance(source, dict):
        return copy.deepcopy(source)
    if isinstance(source, str) and ("\n" in source or ":" in source and not
                                    source.strip().endswith((".yaml", ".yml"))):
        return yaml.load(source, Loader=_UniqueKeyLoader)
    with open(source, "r") as fh:
        return yaml.load(fh, Loader=_UniqueKeyLoader)


class YamlConfig:
    def __init__(self, yaml_file, available_channels, sample_rate=None, verbose=True):
        """yaml_file: path, YAML text or an already-parsed dict."""
        self._yaml_file = yaml_file
        self._sample_rate = sample_rate
        if isinstance(available_channels, str):
            available_channels = [available_channels]
        self._available_channels = list(available_channels)
        self._processing_config = None
        self._read_config()

    def get_config(self, processing_type=None):
        if self._processing_config is None:
            return None
        if processing_type is None:
            return copy.deepcopy(self._processing_config)
        if processing_type not in CONFIGURATION_FIELDS:
            raise ValueError(f'ERROR: Configuration type "{processing_type}" not found!')
        return copy.deepcopy(self._processing_config[processing_type])

    # ------------------------------------------------------------------ parse
    def _read_config(self):
        y = _load(self._yaml_file)
        if not y:
            raise ValueError("ERROR: No configuration loaded. Something went wrong...")
        if "include" in y:
            inc = y.pop("include")
            for f in ([inc] if isinstance(inc, str) else inc):
                y.update(_load(f))
        for old, new in OBSOLETE_KEYS.items():
            y = _rename_keys(y, old, new)

        cfg = {"global": {}}
        for field in CONFIGURATION_FIELDS:
            cfg[field] = {"overall": {}, "channels": {}}
        for p in GLOBAL_PARAMETERS:
            cfg["global"][p] = copy.deepcopy(y.pop(p)) if p in y else None

        for field in CONFIGURATION_FIELDS:
            if field not in y:
                continue
            section = copy.deepcopy(y.pop(field))
            fmap = {"overall": {}, "channels": {}}
            for key, val in section.items():
                if field == "feature" and key in FEATURE_OVERALL:
                    fmap["overall"][key] = val
                elif field == "feature" and key == "global":
                    fmap["overall"].update(val)
                elif isinstance(val, dict):
                    fmap["channels"][key] = val
                else:
                    fmap["overall"][key] = val
            cfg[field] = fmap
        # everything left at top level belongs to feature processing (config.py:205-214)
        for key, val in y.items():
            if key == "global":
                cfg["feature"]["overall"] = copy.deepcopy(val)
            else:
                cfg["feature"]["channels"][key] = copy.deepcopy(val)

        # expand 'all' and comma lists, drop disabled channels (config.py:218-250)
        for field in CONFIGURATION_FIELDS:
            expanded = {}
            for chan, cdict in cfg[field]["channels"].items():
                if isinstance(cdict, dict) and (cdict.get("disable") or
                                                ("run" in cdict and not cdict["run"])):
                    continue
                if chan == "all":
                    for single in self._available_channels:
                        expanded[single] = copy.deepcopy(cdict)
                else:
                    names, _ = utils.split_channel_name(
                        chan, available_channels=self._available_channels, separator=",")
                    for name in names:
                        expanded[name] = copy.deepcopy(cdict)
            cfg[field]["channels"] = expanded

        cfg["feature"] = self._configure_features(cfg["feature"], cfg["global"])
        cfg["trigger"] = self._configure_triggers(cfg["trigger"], cfg["global"])
        cfg["salting"] = self._configure_salting(cfg["salting"], cfg["global"])
        self._processing_config = cfg

    def _merge_global(self, section, global_config, label):
        """Shared head of the trigger / salting passes (config.py:285-320, 329-365): global
        parameters fill the section's ``overall``; every channel needs a dict; returns the copy
        and the list of physical channels its entries name."""
        d = copy.deepcopy(section)
        for k, v in (global_config or {}).items():
            d["overall"].setdefault(k, v)
        split_all = []
        for chan, cc in d["channels"].items():
            if not isinstance(cc, dict):
                raise ValueError(f"ERROR: Channel {chan} has no configuration! Remove "
                                 f"from yaml file or disable it!")
            names, _ = utils.split_channel_name(chan, self._available_channels)
            split_all.extend(names)
        return d, utils.unique_list(split_all)

    def _configure_salting(self, salting_config, global_config):
        d, d["channel_list"] = self._merge_global(salting_config, global_config, "salting")
        return d

    def _configure_triggers(self, trigger_config, global_config):
        """config.py:324-407: an entry either is one trigger (it carries ``run``) or holds one
        dict per trigger algorithm, stored as ``<algorithm>_<trigger_channel>``;
        ``trigger_channel`` (formerly ``trigger_name``) renames the channel."""
        d, channel_list = self._merge_global(trigger_config, global_config, "trigger")
        out = {}
        for chan, cc in d["channels"].items():
            cc = copy.deepcopy(cc)
            trigger_channel = cc.pop("trigger_channel", chan)
            if "run" in cc:
                if not cc["run"]:
                    continue
                cc["channel_name"] = chan
                out[trigger_channel] = cc
                continue
            for algo, ac in cc.items():
                if not isinstance(ac, dict) or "run" not in ac:
                    raise ValueError(f'ERROR: Missing "run" parameter for trigger channel {chan}')
                if not ac["run"]:
                    continue
                ac["channel_name"] = chan
                out[f"{algo}_{trigger_channel}"] = ac
        d["channels"] = out
        d["channel_list"] = channel_list
        return d

    def _length(self, d, kind, current):
        """kind = 'trace' | 'pretrigger' ; samples win over msec (config.py:457-510)."""
        if f"{kind}_length_samples" in d:
            return d[f"{kind}_length_samples"]
        if f"{kind}_length_msec" in d:
            if self._sample_rate is None:
                raise ValueError("ERROR: sample rate is required when trace length "
                                 "is in msec. ")
            return utils.convert_length_msec_to_samples(d[f"{kind}_length_msec"],
                                                        self._sample_rate)
        return current

    def _configure_features(self, feature_config, global_config):
        fd = copy.deepcopy(feature_config)
        for k, v in (global_config or {}).items():
            fd["overall"].setdefault(k, v)

        split_all = []
        for chan in list(fd["channels"].keys()):
            cc = fd["channels"][chan]
            if not isinstance(cc, dict):
                raise ValueError(f"ERROR: Channel {chan} has no configuration! Remove "
                                 f"from yaml file or disable it!")
            names, _ = utils.split_channel_name(chan, self._available_channels)
            split_all.extend(names)
            nb = self._length(fd["overall"], "trace", None)
            npre = self._length(fd["overall"], "pretrigger", None)
            nb = self._length(cc, "trace", nb)
            npre = self._length(cc, "pretrigger", npre)
            if nb is not None and npre is None:
                raise ValueError(f'ERROR: Missing "pretrigger_length_samples" for '
                                 f"channel {chan} !")
            if nb is None and npre is not None:
                raise ValueError(f'ERROR: Missing "trace_length_samples"  for '
                                 f"channel {chan} !")
            algos = []
            for algo in list(cc.keys()):
                ac = cc[algo]
                if not isinstance(ac, dict):
                    continue
                if "run" not in ac:
                    raise ValueError(f'ERROR: Missing "run" parameter for channel {chan}, '
                                     f"algorithm {algo}. Please fix the configuration "
                                     f"yaml file")
                if not ac["run"]:
                    cc.pop(algo)
                    continue
                algos.append(algo)
                ac["nb_samples"] = self._length(ac, "trace", nb)
                ac["nb_pretrigger_samples"] = self._length(ac, "pretrigger", npre)
            if not algos:
                fd["channels"].pop(chan)
            else:
                cc.pop("trace_length_samples", None)
                cc.pop("pretrigger_length_samples", None)

        fd["channel_list"] = utils.unique_list(split_all)
        traces_config, weights = {}, {}
        for chan, cc in fd["channels"].items():
            names, _ = utils.split_channel_name(chan, fd["channel_list"])
            for name in names:
                key = f"weight_{name}"
                if key in cc:
                    weights.setdefault(chan, {})[key] = cc[key]
            for algo, ac in cc.items():
                if not isinstance(ac, dict) or not ac["run"]:
                    continue
                tup = (ac["nb_samples"], ac["nb_pretrigger_samples"])
                traces_config.setdefault(tup, []).extend(names)
        for k in traces_config:
            traces_config[k] = utils.unique_list(traces_config[k])
        fd["traces_config"] = traces_config or None
        fd["weights"] = weights
        return fd
