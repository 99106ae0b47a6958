"""FilterData: in-memory filter-file model with the reference's getters/setters.

Mirrors the part of ``detprocess/core/filterdata.py`` the hot path consumes
(``set_template`` :539-633, ``set_psd`` :636-751, ``get_template`` :450-478,
``get_psd`` :304-377, ``get_csd`` :380-447 single-channel case): storage is
``{channel: {'template_<tag>': array, 'template_<tag>_metadata': dict,
'psd_<tag>': array, 'psd_<tag>_inds': freqs, ...}}``.  The pytesio HDF5 layout
is out of scope (h5py / pytesio are not in this image); ``save_npz`` /
``load_npz`` carry the same dictionary.
"""

import copy

import numpy as np

from .utils import convert_length_msec_to_samples


def _estimate_sampling_rate(freqs):
    f = np.asarray(freqs, dtype=np.float64)
    n = f.shape[-1]
    df = np.abs(f[1] - f[0])
    return float(df * n)


def fold_spectrum(spectrum, fs):
    """Two-sided -> one-sided PSD (positive frequencies, doubled except DC/Nyquist)."""
    s = np.asarray(spectrum)
    n = s.shape[-1]
    k = n // 2 + 1
    f = np.fft.rfftfreq(n, d=1.0 / fs)
    out = np.array(s[..., :k], dtype=np.float64).copy()
    if n % 2:
        out[..., 1:] *= 2.0
    else:
        out[..., 1:-1] *= 2.0
    return f, out


def _channel_name(channels):
    """'a|b' or ['a', 'b'] -> ('a|b', 2): the multi-channel naming of filterdata.py:414-416."""
    if isinstance(channels, str):
        names = [c.strip() for c in channels.split("|")]
    else:
        names = [str(c) for c in channels]
    return "|".join(names), len(names)


class FilterData:
    def __init__(self, verbose=True, filter_data=None):
        self._verbose = verbose
        self._filter_data = filter_data if filter_data is not None else dict()

    # ---------------------------------------------------------------- setters
    def set_template(self, channels, template, sample_rate=None,
                     pretrigger_length_msec=None, pretrigger_length_samples=None,
                     metadata=None, tag="default"):
        if not isinstance(template, np.ndarray):
            raise ValueError('ERROR: "template" argument should be a numpy array!')
        channels, nb_channels = _channel_name(channels)
        if nb_channels == 1 and template.ndim != 1:
            raise ValueError("ERROR: Expecting a 1D array for single channel template")
        if nb_channels > 1 and (template.ndim != 3 or template.shape[0] != nb_channels):
            # filterdata.py:560-576: [nb_channels, nb_templates, nb_samples]
            raise ValueError("ERROR: For multiple channels, expecting 3D array "
                             "[nb channels, nb templates, nb samples]")
        metadata = dict(metadata) if metadata else {}
        if sample_rate is None and "sample_rate" in metadata:
            sample_rate = float(metadata["sample_rate"])
        if sample_rate is None:
            raise ValueError('ERROR: "sample_rate" required!')
        if pretrigger_length_samples is None:
            if pretrigger_length_msec is None:
                raise ValueError("ERROR: pretrigger length (samples or msec) required!")
            pretrigger_length_samples = convert_length_msec_to_samples(
                pretrigger_length_msec, sample_rate)
        metadata.update(sample_rate=float(sample_rate),
                        nb_samples=int(template.shape[-1]),
                        nb_pretrigger_samples=int(pretrigger_length_samples),
                        channel=channels)
        ch = self._filter_data.setdefault(channels, dict())
        ch[f"template_{tag}"] = np.array(template, dtype=np.float64)
        ch[f"template_{tag}_inds"] = (np.arange(template.shape[-1])
                                      - pretrigger_length_samples) / float(sample_rate)
        ch[f"template_{tag}_metadata"] = metadata

    def set_psd(self, channels, psd, psd_freqs, sample_rate=None,
                pretrigger_length_msec=None, pretrigger_length_samples=None,
                metadata=None, tag="default"):
        psd = np.asarray(psd, dtype=np.float64)
        psd_freqs = np.asarray(psd_freqs, dtype=np.float64)
        if not isinstance(channels, str):
            raise ValueError("ERROR: only single-channel PSDs are supported here")
        if psd.ndim != 1 or psd_freqs.shape != psd.shape:
            raise ValueError("ERROR: psd shape is not consistent with number of channels")
        if not np.any(psd_freqs < 0):
            raise ValueError("ERROR: psd needs to be two-sided!")   # filterdata.py:673-676
        metadata = dict(metadata) if metadata else {}
        fs_arr = _estimate_sampling_rate(psd_freqs)
        if sample_rate is None:
            sample_rate = float(metadata.get("sample_rate", fs_arr))
        elif round(fs_arr) != round(sample_rate):
            raise ValueError("ERROR: sample_rate is inconsistent with frequency array!")
        metadata.update(sample_rate=float(sample_rate), nb_samples=int(psd.shape[-1]),
                        channel=channels)
        if pretrigger_length_samples is not None:
            metadata["nb_pretrigger_samples"] = int(pretrigger_length_samples)
        elif pretrigger_length_msec is not None:
            metadata["nb_pretrigger_samples"] = convert_length_msec_to_samples(
                pretrigger_length_msec, sample_rate)
        ch = self._filter_data.setdefault(channels, dict())
        ch[f"psd_{tag}"] = psd.copy()
        ch[f"psd_{tag}_inds"] = psd_freqs.copy()
        ch[f"psd_{tag}_metadata"] = metadata

    def set_csd(self, channels, csd, csd_freqs, sample_rate=None, metadata=None, tag="default",
                **kwargs):
        """Single-channel CSD == PSD (get_csd falls back to get_psd, filterdata.py:417-420);
        an ``a|b`` channel stores the two-sided [nb_channels, nb_channels, nb_samples] array."""
        csd = np.asarray(csd)
        channels, nb_channels = _channel_name(channels)
        if nb_channels == 1:
            if csd.ndim == 3 and csd.shape[0] == 1 and csd.shape[1] == 1:
                csd = csd[0, 0]
            self.set_psd(channels, np.real(csd), csd_freqs, sample_rate=sample_rate,
                         metadata=metadata, tag=tag, **kwargs)
            return
        csd_freqs = np.asarray(csd_freqs, dtype=np.float64)
        if csd.ndim != 3 or csd.shape[0] != nb_channels or csd.shape[1] != nb_channels \
                or csd_freqs.shape != csd.shape[-1:]:
            raise ValueError("ERROR: csd shape is not consistent with number of channels")
        if not np.any(csd_freqs < 0):
            raise ValueError("ERROR: csd needs to be two-sided!")
        metadata = dict(metadata) if metadata else {}
        fs_arr = _estimate_sampling_rate(csd_freqs)
        if sample_rate is None:
            sample_rate = float(metadata.get("sample_rate", fs_arr))
        elif round(fs_arr) != round(sample_rate):
            raise ValueError("ERROR: sample_rate is inconsistent with frequency array!")
        metadata.update(sample_rate=float(sample_rate), nb_samples=int(csd.shape[-1]),
                        channel=channels)
        ch = self._filter_data.setdefault(channels, dict())
        ch[f"csd_{tag}"] = np.array(csd, dtype=np.complex128)
        ch[f"csd_{tag}_inds"] = csd_freqs.copy()
        ch[f"csd_{tag}_metadata"] = metadata

    # ---------------------------------------------------------------- getters
    def _get_param_array(self, param_name, channel, tag="default", return_metadata=False):
        if channel not in self._filter_data:
            msg = f'ERROR: Channel "{channel}" not available!'
            if self._filter_data:
                msg += " List of channels in filter file: " + str(list(self._filter_data))
            raise ValueError(msg)
        name = f"{param_name}_{tag}"
        if name not in self._filter_data[channel]:
            raise ValueError(f"ERROR: Parameter {name} not found for channel {channel}!")
        vals = self._filter_data[channel][name].copy()
        inds = self._filter_data[channel].get(name + "_inds")
        meta = copy.deepcopy(self._filter_data[channel].get(name + "_metadata", {}))
        return (vals, inds, meta) if return_metadata else (vals, inds)

    def get_template(self, channel, tag="default", return_metadata=False):
        return self._get_param_array("template", channel, tag, return_metadata)

    def get_psd(self, channels, tag="default", fold=False, return_metadata=False):
        psd, freqs, meta = self._get_param_array("psd", channels, tag, True)
        if fold:
            freqs, psd = fold_spectrum(psd, float(meta.get("sample_rate",
                                                           _estimate_sampling_rate(freqs))))
        return (psd, freqs, meta) if return_metadata else (psd, freqs)

    def get_csd(self, channels, tag="default", fold=False, return_metadata=False):
        channels, nb_channels = _channel_name(channels)
        if nb_channels == 1:
            return self.get_psd(channels, tag=tag, fold=fold, return_metadata=return_metadata)
        csd, freqs, meta = self._get_param_array("csd", channels, tag, True)   # filterdata.py:422-428
        if fold:
            fs = float(meta.get("sample_rate", _estimate_sampling_rate(freqs)))
            n = csd.shape[-1]
            k = n // 2 + 1
            freqs = np.fft.rfftfreq(n, d=1.0 / fs)
            csd = csd[..., :k].copy()
            csd[..., 1:(k if n % 2 else k - 1)] *= 2.0
        return (csd, freqs, meta) if return_metadata else (csd, freqs)

    def describe(self):
        for chan, d in self._filter_data.items():
            print(f"Channel {chan}:")
            for k, v in d.items():
                if not k.endswith(("_metadata", "_inds")):
                    print(f"   {k}: {getattr(v, 'shape', '')}")

    # ------------------------------------------------------------------ HDF5
    @staticmethod
    def _h5py():
        try:
            import h5py
        except ImportError as exc:
            raise ImportError(
                "ERROR: h5py is required for HDF5 filter files (detprocess writes them through "
                "pytesio's FilterH5IO, filterdata.py:218-246, 270-300).  It is not installed in "
                "this environment: use save_npz / load_npz, or install h5py.") from exc
        return h5py

    def load_hdf5(self, file_name, overwrite=True):
        """filterdata.py:218-246.  Layout read: one group per channel; a parameter is either a
        dataset (array, its attributes = the ``<name>_metadata`` dictionary, with an optional
        sibling dataset ``<name>_inds``) or a pandas fixed-format node (a group holding
        ``values`` and ``index`` datasets -- how a 1-D template / psd stored as ``pd.Series``
        lands in the file): the index becomes ``<name>_inds``.  The authoritative layout lives
        in pytesio (absent from the reference tree and from this image), so this reader is
        checked against files written by ``save_hdf5`` only."""
        h5py = self._h5py()
        data = {}
        with h5py.File(file_name, "r") as f:
            for chan in f:
                grp = f[chan]
                if not isinstance(grp, h5py.Group):
                    continue
                d = data.setdefault(chan, {})
                for name, node in grp.items():
                    meta = {k: (v.item() if hasattr(v, "item") and np.ndim(v) == 0 else v)
                            for k, v in node.attrs.items()
                            if not k.startswith(("pandas_", "CLASS", "VERSION", "TITLE"))}
                    if isinstance(node, h5py.Group):
                        if "values" not in node:
                            continue
                        d[name] = np.asarray(node["values"])
                        if "index" in node:
                            d[name + "_inds"] = np.asarray(node["index"])
                    else:
                        d[name] = np.asarray(node)
                    if meta and not name.endswith("_inds"):
                        d[name + "_metadata"] = meta
        self.set_data(data, overwrite=overwrite)

    def save_hdf5(self, file_name, overwrite=False):
        """filterdata.py:270-300: one group per channel, one dataset per array, metadata as
        attributes (see load_hdf5 for the caveat on the layout)."""
        h5py = self._h5py()
        with h5py.File(file_name, "a") as f:
            for chan, d in self._filter_data.items():
                grp = f.require_group(chan)
                for name, val in d.items():
                    if name.endswith("_metadata"):
                        continue
                    if name in grp:
                        if not overwrite:
                            continue
                        del grp[name]
                    ds = grp.create_dataset(name, data=np.asarray(val))
                    for k, v in d.get(name + "_metadata", {}).items():
                        if v is not None:
                            ds.attrs[k] = v

    def set_data(self, data, overwrite=False):
        """filterdata.py:248-268."""
        if not isinstance(data, dict):
            raise ValueError("ERROR: filter data should be a dictionary!")
        for key, item in data.items():
            if key not in self._filter_data:
                self._filter_data[key] = item
                continue
            for par_name, value in item.items():
                if overwrite or par_name not in self._filter_data[key]:
                    self._filter_data[key][par_name] = value

    # -------------------------------------------------------------- npz carry
    def save_npz(self, file_name):
        flat = {}
        for chan, d in self._filter_data.items():
            for k, v in d.items():
                if k.endswith("_metadata"):
                    for mk, mv in v.items():
                        flat[f"{chan}::{k}::{mk}"] = np.asarray(mv)
                else:
                    flat[f"{chan}::{k}"] = v
        np.savez_compressed(file_name, **flat)

    def load_npz(self, file_name, overwrite=True):
        z = np.load(file_name, allow_pickle=False)
        for key in z.files:
            parts = key.split("::")
            ch = self._filter_data.setdefault(parts[0], dict())
            if len(parts) == 2:
                if overwrite or parts[1] not in ch:
                    ch[parts[1]] = z[key]
            else:
                v = z[key]
                ch.setdefault(parts[1], dict())[parts[2]] = v.item() if v.ndim == 0 else v
