"""N-channel x M-template optimal filter: host precompute and the handle on ``ofx_nxm``.

``build_nxm_filter`` is the one-time setup the reference does in
``ProcessingData.instantiate_OF_base`` for an ``a|b`` channel (processing_data.py:294-381:
``FilterData.get_csd(..., fold=False)`` -> ``OFBase.set_csd`` with AC coupling and notches,
``get_template`` -> ``add_template``, ``calc_phi``): template FFTs, the inverse CSD per
frequency bin, phi = S^H C^-1 and the M x M weight matrix, in fp64 with NumPy.  ``NxMPlan``
runs the per-event work (``qp.OFnxm(...).calc()``, ``get_fit_withdelay`` / ``get_fit_nodelay``:
algorithms.py:241-262) for a whole batch on the GPU; there is no CPU fallback.
"""

import ctypes as C
from math import ceil, floor

import numpy as np

from . import _lib
from .filters import apply_coupling_and_notches


class NxMTables:
    """phi [M, C, K], icov [C, C, K] (one-sided, K = N/2 + 1), pinv [M, M], all fp64."""

    def __init__(self, phi, icov, pinv, n_samples, n_pretrigger, fs):
        self.phi, self.icov, self.pinv = phi, icov, pinv
        self.n_tmpl, self.n_chan = phi.shape[0], phi.shape[1]
        self.n_samples, self.n_pretrigger, self.fs = int(n_samples), int(n_pretrigger), float(fs)
        self.ampres = np.sqrt(np.diag(pinv))


def _dropped_bins(n, fs, coupling, ignored_frequency_peaks, ignore_harmonics):
    """Bins whose CSD OFBase.set_csd sends to infinity (same rule as the 1x1 PSD edits)."""
    return ~np.isfinite(apply_coupling_and_notches(np.ones(n), fs, coupling,
                                                   ignored_frequency_peaks, ignore_harmonics))


def build_nxm_filter(templates, csd, fs, n_pretrigger, coupling="AC",
                     ignored_frequency_peaks=None, ignore_harmonics=False):
    """templates [C, M, N] (amps, any normalisation), csd [C, C, N] two-sided A^2/Hz in
    fftfreq order (``FilterData.get_csd(..., fold=False)``)."""
    s = np.asarray(templates, dtype=np.float64)
    c = np.asarray(csd, dtype=np.complex128)
    if s.ndim != 3 or c.ndim != 3 or c.shape[0] != c.shape[1]:
        raise ValueError("ERROR: templates must be [n_channels, n_templates, samples] and the "
                         "csd [n_channels, n_channels, samples]")
    if c.shape[0] != s.shape[0]:
        raise ValueError(f"ERROR: {s.shape[0]} template channels but a {c.shape[0]}-channel csd")
    n = s.shape[2]
    if c.shape[2] != n:
        raise ValueError(f"ERROR: Number of samples is not consistent between template (={n}) "
                         f"and csd (={c.shape[2]})")
    if n % 2:
        raise ValueError("ERROR: the NxM engine needs an even number of samples")
    k = n // 2 + 1
    drop = _dropped_bins(n, fs, coupling, ignored_frequency_peaks, ignore_harmonics)[:k]
    ck = np.ascontiguousarray(np.moveaxis(c[:, :, :k], 2, 0))        # [k, a, b]
    icov = np.zeros_like(ck)
    icov[~drop] = np.linalg.inv(ck[~drop])
    sf = np.fft.fft(s, axis=-1)[:, :, :k]                            # [a, m, k]
    phi = np.einsum("amk,kab->mbk", np.conj(sf), icov)               # [m, b, k]
    w = np.full(k, 2.0)
    w[0] = w[-1] = 1.0
    p = np.einsum("mbk,bnk,k->mn", phi, sf, w).real / (n * fs)
    p = 0.5 * (p + p.T)
    return NxMTables(np.ascontiguousarray(phi), np.ascontiguousarray(np.moveaxis(icov, 0, 2)),
                     np.ascontiguousarray(np.linalg.inv(p)), n, n_pretrigger, fs)


def nxm_search_range(n_samples, n_pretrigger, fs, window_min_from_trig_usec=None,
                     window_max_from_trig_usec=None, window_min_index=None,
                     window_max_index=None):
    """Half-open rolled range of ``get_fit_withdelay`` (same rule as the of1x1 delay fit:
    the usec arguments win over the indices, floor / ceil, clipped to the trace)."""
    lo = hi = None
    if window_min_from_trig_usec is not None:
        lo = floor(n_pretrigger + window_min_from_trig_usec * fs * 1e-6)
    elif window_min_index is not None:
        lo = int(window_min_index)
    if window_max_from_trig_usec is not None:
        hi = ceil(n_pretrigger + window_max_from_trig_usec * fs * 1e-6)
    elif window_max_index is not None:
        hi = int(window_max_index)
    if lo is None or lo < 0:
        lo = 0
    if hi is None or hi > n_samples:
        hi = n_samples
    return int(lo), int(hi)


class NxMPlan:
    """Handle on an ``ofx_nxm`` object (include/ofx.h)."""

    def __init__(self, tables: NxMTables, max_batch=4096, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.tables = tables
        self.n_samples, self.n_pretrigger, self.fs = tables.n_samples, tables.n_pretrigger, tables.fs
        self.n_chan, self.n_tmpl = tables.n_chan, tables.n_tmpl
        self.n_channels_total = self.n_chan
        self.device = int(device)
        _lib.check(self._lib.ofx_nxm_create(C.byref(self._h), self.n_samples, self.n_pretrigger,
                                            self.fs, self.n_chan, self.n_tmpl, int(max_batch),
                                            self.device), "ofx_nxm_create")
        phi = np.ascontiguousarray(tables.phi, dtype=np.complex128)
        icov = np.ascontiguousarray(tables.icov, dtype=np.complex128)
        pinv = np.ascontiguousarray(tables.pinv, dtype=np.float64)
        _lib.check(self._lib.ofx_nxm_set_filter(self._h, phi.ctypes.data, icov.ctypes.data,
                                                pinv.ctypes.data), "ofx_nxm_set_filter")
        self.searches = []

    def add_search(self, kind, lo=0, hi=None, outside=False, interpolate=False):
        """kind 'nodelay' or 'delay' over rolled bins [lo, hi); interpolate: the delay fit's
        interpolate_t0 (algorithms.py:152, 259); returns the search id."""
        k = {"nodelay": _lib.SEARCH_NODELAY, "delay": _lib.SEARCH_DELAY}[kind]
        if interpolate and kind == "delay":
            k = _lib.SEARCH_DELAY_INTERP
        hi = self.n_samples if hi is None else hi
        sid = self._lib.ofx_nxm_add_search(self._h, k, int(lo), int(hi), int(bool(outside)))
        if sid < 0:
            _lib.check(-sid, "ofx_nxm_add_search")
        self.searches.append((kind, int(lo), int(hi), bool(outside)))
        return sid

    def reset_searches(self):
        _lib.check(self._lib.ofx_nxm_reset_searches(self._h), "ofx_nxm_reset_searches")
        self.searches = []

    def set_channels(self, n_channels_total, index):
        idx = np.ascontiguousarray(index, dtype=np.int32)
        if idx.shape != (self.n_chan,):
            raise ValueError(f"ERROR: expected {self.n_chan} channel indices")
        _lib.check(self._lib.ofx_nxm_set_channels(self._h, int(n_channels_total), idx.ctypes.data),
                   "ofx_nxm_set_channels")
        self.n_channels_total = int(n_channels_total)

    @property
    def row_floats(self):
        return self._lib.ofx_nxm_row_floats(self._h)

    def record(self, out, sid):
        """Split search ``sid`` of a result matrix: amps [B, M], t0, chi2, index."""
        o = sid * (self.n_tmpl + 3)
        m = self.n_tmpl
        return out[:, o:o + m], out[:, o + m], out[:, o + m + 1], out[:, o + m + 2]

    def _batch(self, shape):
        want = (self.n_channels_total, self.n_samples)
        if len(shape) != 3 or tuple(shape[1:]) != want:
            raise ValueError(f"ERROR: events must be [B, {want[0]}, {want[1]}], got {tuple(shape)}")
        return int(shape[0])

    def process(self, events, valid=None):
        """events float32 [B, n_channels_total, N]: NumPy (staged over PCIe) or a CUDA tensor
        (in place, asynchronous on the current stream).  Returns [B, row_floats] float32."""
        row = self.row_floats
        if isinstance(events, np.ndarray):
            ev = np.ascontiguousarray(events, dtype=np.float32)
            b = self._batch(ev.shape)
            out = np.empty((b, row), dtype=np.float32)
            v_ptr = None
            if valid is not None:
                valid = np.ascontiguousarray(valid, dtype=np.uint8)
                v_ptr = valid.ctypes.data
            _lib.check(self._lib.ofx_nxm_process(self._h, ev.ctypes.data, v_ptr, b, _lib.MEM_HOST,
                                                 out.ctypes.data, _lib.MEM_HOST, None),
                       "ofx_nxm_process")
            return out
        import torch
        if not (isinstance(events, torch.Tensor) and events.is_cuda
                and events.dtype == torch.float32):
            raise TypeError("events must be a float32 NumPy array or CUDA tensor")
        if events.device.index != self.device:
            raise ValueError(f"events live on cuda:{events.device.index}, plan on "
                             f"cuda:{self.device}")
        ev = events.contiguous()
        b = self._batch(tuple(ev.shape))
        out = torch.empty((b, row), dtype=torch.float32, device=ev.device)
        v_ptr = None
        if valid is not None:
            valid = torch.as_tensor(valid).to(device=ev.device, dtype=torch.uint8).contiguous()
            v_ptr = valid.data_ptr()
        stream = torch.cuda.current_stream(ev.device).cuda_stream
        _lib.check(self._lib.ofx_nxm_process(self._h, ev.data_ptr(), v_ptr, b, _lib.MEM_DEVICE,
                                             out.data_ptr(), _lib.MEM_DEVICE, C.c_void_p(stream)),
                   "ofx_nxm_process")
        return out

    def process_adc(self, adc, trigger_index, scale, offset):
        """Events cut on the GPU out of continuous int16 streams ``adc [n_channels_total,
        n_stream]`` (NumPy or CUDA tensor) around ``trigger_index [B]``, then fitted; scale /
        offset per channel as in ``OFPlan.process_adc``.  Returns float32 [B, row_floats] of
        the same kind as ``adc``."""
        trig = np.ascontiguousarray(trigger_index, dtype=np.int64)
        b = int(trig.shape[0])
        ct = self.n_channels_total
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (ct,)))
        of = np.ascontiguousarray(np.broadcast_to(np.asarray(offset, dtype=np.float64), (ct,)))
        row = self.row_floats
        if isinstance(adc, np.ndarray):
            a = np.ascontiguousarray(adc, dtype=np.int16)
            if a.ndim != 2 or a.shape[0] != ct:
                raise ValueError(f"ERROR: adc must be [{ct}, n_stream], got {a.shape}")
            out = np.empty((b, row), dtype=np.float32)
            _lib.check(self._lib.ofx_nxm_process_adc(
                self._h, a.ctypes.data, int(a.shape[1]), _lib.MEM_HOST, trig.ctypes.data, b,
                sc.ctypes.data, of.ctypes.data, out.ctypes.data, _lib.MEM_HOST, None),
                "ofx_nxm_process_adc")
            return out
        import torch
        if not (isinstance(adc, torch.Tensor) and adc.is_cuda and adc.dtype == torch.int16):
            raise TypeError("adc must be an int16 NumPy array or CUDA tensor")
        a = adc.contiguous()
        if a.dim() != 2 or a.shape[0] != ct:
            raise ValueError(f"ERROR: adc must be [{ct}, n_stream], got {tuple(a.shape)}")
        out = torch.empty((b, row), dtype=torch.float32, device=a.device)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _lib.check(self._lib.ofx_nxm_process_adc(
            self._h, a.data_ptr(), int(a.shape[1]), _lib.MEM_DEVICE, trig.ctypes.data, b,
            sc.ctypes.data, of.ctypes.data, out.data_ptr(), _lib.MEM_DEVICE, C.c_void_p(stream)),
            "ofx_nxm_process_adc")
        return out

    def close(self):
        if self._h:
            self._lib.ofx_nxm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
