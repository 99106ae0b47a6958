"""OFPlan: Python handle on an ``ofx_plan`` (include/ofx.h).

One plan = one ``(nb_samples, nb_pretrigger_samples)`` key of the reference's
``_OF_base_objs`` dictionary (processing_data.py:274-286): a set of filter
slots (template_tag x csd_tag), the of1x1 searches registered on each slot, the
time-domain windows, and the channel algebra applied on load.  ``process``
runs the whole hot path for a batch of events on the GPU and returns the
feature matrix; it is the only compute entry and it has no CPU fallback.
"""

import ctypes as C

import numpy as np

from . import _lib
from .filters import FilterTables

_ENGINES = {"auto": _lib.ENGINE_AUTO, "fused": _lib.ENGINE_FUSED,
            "rocfft": _lib.ENGINE_ROCFFT, "lds": _lib.ENGINE_LDS}


def _torch():
    import torch
    return torch


class OFPlan:
    def __init__(self, n_samples, n_pretrigger, fs, max_batch=4096, device=0,
                 engine="auto"):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.n_samples = int(n_samples)
        self.n_pretrigger = int(n_pretrigger)
        self.fs = float(fs)
        self.device = int(device)
        self.n_channels = 1
        _lib.check(self._lib.ofx_plan_create(C.byref(self._h), self.n_samples,
                                             self.n_pretrigger, self.fs,
                                             int(max_batch), self.device,
                                             _ENGINES[engine]), "ofx_plan_create")
        self.filters = {}

    # ------------------------------------------------------------------ config
    @property
    def engine(self):
        e = self._lib.ofx_plan_engine(self._h)
        return {v: k for k, v in _ENGINES.items()}[e]

    def set_filter(self, slot, tables: FilterTables):
        if tables.n_samples != self.n_samples:
            raise ValueError(f"ERROR: Number of samples is not consistent between "
                             f"raw data (={self.n_samples}) and filter "
                             f"(={tables.n_samples})")
        wf = np.ascontiguousarray(tables.wf, dtype=np.complex128)
        g = np.ascontiguousarray(tables.g, dtype=np.float64)
        s = np.ascontiguousarray(tables.s, dtype=np.complex128)
        _lib.check(self._lib.ofx_plan_set_filter(
            self._h, int(slot), wf.ctypes.data, g.ctypes.data, s.ctypes.data,
            float(tables.norm), float(tables.tres_sum)), "ofx_plan_set_filter")
        self.filters[int(slot)] = tables

    def add_search(self, slot, kind, lo=0, hi=None, outside=False,
                   lowchi2_fcutoff=10000.0, interpolate=False):
        kind_i = {"nodelay": _lib.SEARCH_NODELAY, "delay": _lib.SEARCH_DELAY}[kind]
        if interpolate and kind == "delay":
            kind_i = _lib.SEARCH_DELAY_INTERP
        if hi is None:
            hi = self.n_samples
        sid = self._lib.ofx_plan_add_search(self._h, int(slot), kind_i, int(lo),
                                            int(hi), int(bool(outside)),
                                            float(lowchi2_fcutoff))
        if sid < 0:
            _lib.check(-sid, "ofx_plan_add_search")
        return sid

    def add_tdwindow(self, lo, hi):
        wid = self._lib.ofx_plan_add_tdwindow(self._h, int(lo), int(hi))
        if wid < 0:
            _lib.check(-wid, "ofx_plan_add_tdwindow")
        return wid

    def add_band(self, k_lo, k_hi):
        """psd_amp band over one-sided FFT bins [k_lo, k_hi)."""
        bid = self._lib.ofx_plan_add_band(self._h, int(k_lo), int(k_hi))
        if bid < 0:
            _lib.check(-bid, "ofx_plan_add_band")
        return bid

    def set_channels(self, n_channels, chan_index, weights=None):
        idx = np.ascontiguousarray(chan_index, dtype=np.int32)
        w = np.ones(len(idx)) if weights is None else np.ascontiguousarray(
            weights, dtype=np.float64)
        _lib.check(self._lib.ofx_plan_set_channels(self._h, int(n_channels), len(idx),
                                                   idx.ctypes.data, w.ctypes.data),
                   "ofx_plan_set_channels")
        self.n_channels = int(n_channels)

    def reset(self):
        _lib.check(self._lib.ofx_plan_reset(self._h), "ofx_plan_reset")
        self.filters = {}
        self.n_channels = 1

    @property
    def row_floats(self):
        return self._lib.ofx_plan_row_floats(self._h)

    def search_offset(self, slot, search):
        return self._lib.ofx_plan_search_offset(self._h, int(slot), int(search))

    def tdwindow_offset(self, window):
        return self._lib.ofx_plan_tdwindow_offset(self._h, int(window))

    def band_offset(self, band):
        return self._lib.ofx_plan_band_offset(self._h, int(band))

    # ----------------------------------------------------------------- timing
    def enable_timing(self, on=True):
        _lib.check(self._lib.ofx_plan_enable_timing(self._h, int(on)), "enable_timing")

    def kernel_time(self):
        ms = C.c_double()
        n = C.c_longlong()
        _lib.check(self._lib.ofx_plan_kernel_time(self._h, C.byref(ms), C.byref(n)),
                   "ofx_plan_kernel_time")
        return ms.value, n.value

    # ---------------------------------------------------------------- compute
    def process(self, traces, valid=None, out=None):
        """Run the hot path on ``traces``.

        traces: float32, shape [B, N] or [B, C, N]; a CUDA ``torch.Tensor``
        (zero-copy, asynchronous on the current stream) or a NumPy array / CPU
        tensor (staged over PCIe chunk by chunk).  valid: optional bool/uint8
        [B]; rows with 0 come back as -999999.0.  Returns [B, row_floats]
        float32 of the same kind as ``traces`` (or fills ``out``).
        """
        row = self.row_floats
        is_np = isinstance(traces, np.ndarray)
        if is_np:
            if traces.dtype != np.float32 or not traces.flags["C_CONTIGUOUS"]:
                traces = np.ascontiguousarray(traces, dtype=np.float32)
            B = self._check_shape(traces.shape)
            if out is None:
                out = np.empty((B, row), dtype=np.float32)
            v_ptr = None
            if valid is not None:
                valid = np.ascontiguousarray(valid, dtype=np.uint8)
                v_ptr = valid.ctypes.data
            _lib.check(self._lib.ofx_process(self._h, traces.ctypes.data, v_ptr, B,
                                             _lib.MEM_HOST, out.ctypes.data,
                                             _lib.MEM_HOST, None), "ofx_process")
            return out
        torch = _torch()
        if not isinstance(traces, torch.Tensor):
            raise TypeError("traces must be a numpy array or a torch tensor")
        if not traces.is_cuda:
            res = self.process(traces.numpy(), None if valid is None else
                               np.asarray(valid), None)
            return torch.from_numpy(res)
        if traces.dtype != torch.float32:
            raise TypeError("device traces must be float32")
        if traces.device.index != self.device:
            raise ValueError(f"traces live on cuda:{traces.device.index}, plan on "
                             f"cuda:{self.device}")
        traces = traces.contiguous()
        B = self._check_shape(tuple(traces.shape))
        if out is None:
            out = torch.empty((B, row), dtype=torch.float32, device=traces.device)
        v_ptr = None
        if valid is not None:
            valid = torch.as_tensor(valid).to(device=traces.device, dtype=torch.uint8).contiguous()
            v_ptr = valid.data_ptr()
        stream = torch.cuda.current_stream(traces.device).cuda_stream
        _lib.check(self._lib.ofx_process(self._h, traces.data_ptr(), v_ptr, B,
                                         _lib.MEM_DEVICE, out.data_ptr(),
                                         _lib.MEM_DEVICE, C.c_void_p(stream)),
                   "ofx_process")
        return out

    def process_adc(self, adc, trigger_index, scale, offset):
        """Cut events out of continuous raw-data streams and run the hot path on them.

        adc: int16 [C, n_stream] (C = the plan's channel count), a NumPy array or a CUDA
        tensor; trigger_index: int64 [B] sample index of each trigger in the stream;
        scale, offset: per channel, amps = float32(adc * scale) + offset (pytesio's
        adctoamp conversion, coefficients from the file's detector_config).  Event b is
        stream[:, trigger_index[b] - n_pretrigger : ... + n_samples]; a window that does not
        fit gives a row of -999999.0 (processing_data.py:640-656).  Returns float32
        [B, row_floats] of the same kind as ``adc``.
        """
        trig = np.ascontiguousarray(trigger_index, dtype=np.int64)
        B = int(trig.shape[0])
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64),
                                                  (self.n_channels,)))
        of = np.ascontiguousarray(np.broadcast_to(np.asarray(offset, dtype=np.float64),
                                                  (self.n_channels,)))
        row = self.row_floats
        if isinstance(adc, np.ndarray):
            a = np.ascontiguousarray(adc, dtype=np.int16)
            if a.ndim == 1:
                a = a[None, :]
            if a.shape[0] != self.n_channels:
                raise ValueError(f"ERROR: adc has {a.shape[0]} channels, plan expects "
                                 f"{self.n_channels}")
            out = np.empty((B, row), dtype=np.float32)
            _lib.check(self._lib.ofx_process_adc(
                self._h, a.ctypes.data, int(a.shape[1]), _lib.MEM_HOST, trig.ctypes.data, B,
                sc.ctypes.data, of.ctypes.data, out.ctypes.data, _lib.MEM_HOST, None),
                "ofx_process_adc")
            return out
        torch = _torch()
        if not (isinstance(adc, torch.Tensor) and adc.is_cuda and adc.dtype == torch.int16):
            raise TypeError("adc must be an int16 NumPy array or CUDA tensor")
        a = adc.contiguous()
        if a.dim() == 1:
            a = a[None, :]
        if a.shape[0] != self.n_channels:
            raise ValueError(f"ERROR: adc has {a.shape[0]} channels, plan expects "
                             f"{self.n_channels}")
        out = torch.empty((B, row), dtype=torch.float32, device=a.device)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _lib.check(self._lib.ofx_process_adc(
            self._h, a.data_ptr(), int(a.shape[1]), _lib.MEM_DEVICE, trig.ctypes.data, B,
            sc.ctypes.data, of.ctypes.data, out.data_ptr(), _lib.MEM_DEVICE,
            C.c_void_p(stream)), "ofx_process_adc")
        return out

    def _check_shape(self, shape):
        if len(shape) == 2:
            if self.n_channels != 1:
                raise ValueError("ERROR: traces must be [B, C, N] when channels are set")
            B, n = shape
        elif len(shape) == 3:
            B, c, n = shape
            if c != self.n_channels:
                raise ValueError(f"ERROR: traces have {c} channels, plan expects "
                                 f"{self.n_channels}")
        else:
            raise ValueError("ERROR: traces must be [B, N] or [B, C, N]")
        if n != self.n_samples:
            raise ValueError(f"ERROR: Number of samples is not consistent between "
                             f"raw data (={n}) and plan (={self.n_samples})")
        return int(B)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ofx_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_traces(n_traces, n_samples, template, sigma, amp_lo, amp_hi,
                 pulse_fraction=0.5, max_delay=2000, seed=0, first_index=0,
                 device=0, out=None, return_truth=True, psd=None, fs=None):
    """Device-side synthetic events: returns (traces, truth).

    psd=None: white noise of standard deviation ``sigma`` (ofx_synth_traces).
    psd=two-sided PSD J (A^2/Hz, fftfreq order) with ``fs``: coloured Gaussian noise
    irfft(sqrt(J N fs / 2) xi) as SURVEY.md section 8d prescribes (ofx_synth_traces_psd).
    """
    torch = _torch()
    lib = _lib.load()
    dev = torch.device("cuda", device)
    if out is None:
        out = torch.empty((n_traces, n_samples), dtype=torch.float32, device=dev)
    truth = torch.empty((n_traces, 2), dtype=torch.float32, device=dev) \
        if return_truth else None
    t = torch.as_tensor(np.asarray(template, dtype=np.float32), device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if psd is not None:
        K = n_samples // 2 + 1
        J1 = np.asarray(psd, dtype=np.float64)[:K]
        namp = torch.as_tensor((np.sqrt(J1 * n_samples * float(fs) / 2.0) / n_samples)
                               .astype(np.float32), device=dev)
        _lib.check(lib.ofx_synth_traces_psd(
            out.data_ptr(), truth.data_ptr() if truth is not None else None, int(n_traces),
            int(first_index), int(n_samples), t.data_ptr(), namp.data_ptr(), float(amp_lo),
            float(amp_hi), float(pulse_fraction), int(max_delay), int(seed),
            C.c_void_p(stream)), "ofx_synth_traces_psd")
        torch.cuda.current_stream(dev).synchronize()
        return out, truth
    _lib.check(lib.ofx_synth_traces(out.data_ptr(),
                                    truth.data_ptr() if truth is not None else None,
                                    int(n_traces), int(first_index), int(n_samples),
                                    t.data_ptr(), float(sigma), float(amp_lo),
                                    float(amp_hi), float(pulse_fraction),
                                    int(max_delay), int(seed), C.c_void_p(stream)),
               "ofx_synth_traces")
    torch.cuda.current_stream(dev).synchronize()   # t must outlive the kernel
    return out, truth


class SynthSource:
    """Device-side generator of the bench events (SURVEY.md section 8d recipe), keyed by
    (seed, global event index): ``fill(lo, hi, buf)`` writes the events [lo, hi) into
    buf[: hi - lo] on the current stream without a host synchronisation (the tables live as
    long as the object), so it can run on a producer stream beside the hot path
    (``detprocess_amd.dist.run_sharded``)."""

    def __init__(self, n_samples, template, psd, fs, amp_lo, amp_hi, pulse_fraction=0.5,
                 max_delay=2000, seed=0, device=0, white=False):
        """white=True: white Gaussian noise of the PSD's median level instead of noise coloured by
        J (ofx_synth_traces: one HBM write per trace and nothing else -- the source of the streamed
        form of the bench, which must not be slower than the path it feeds; the coloured form,
        k_spectrum -> rocFFT C2R -> k_add_pulse, moves ~5x the trace bytes)."""
        torch = _torch()
        self.white = bool(white)
        self._sigma = float(np.sqrt(np.median(np.asarray(psd, dtype=np.float64)) * float(fs)))
        self._lib = _lib.load()
        self.n_samples = int(n_samples)
        self.device = torch.device("cuda", device)
        self._t = torch.as_tensor(np.asarray(template, dtype=np.float32), device=self.device)
        K = self.n_samples // 2 + 1
        J1 = np.asarray(psd, dtype=np.float64)[:K]
        self._namp = torch.as_tensor(
            (np.sqrt(J1 * self.n_samples * float(fs) / 2.0) / self.n_samples).astype(np.float32),
            device=self.device)
        self._args = (float(amp_lo), float(amp_hi), float(pulse_fraction), int(max_delay),
                      int(seed))

    def fill(self, lo, hi, buf):
        torch = _torch()
        n = int(hi) - int(lo)
        if n <= 0:
            return
        if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.shape[0] < n:
            raise ValueError("ERROR: SynthSource.fill needs a contiguous float32 buffer of at "
                             "least hi - lo events")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        a_lo, a_hi, frac, dmax, seed = self._args
        if self.white:
            _lib.check(self._lib.ofx_synth_traces(
                buf.data_ptr(), None, n, int(lo), self.n_samples, self._t.data_ptr(), self._sigma,
                a_lo, a_hi, frac, dmax, seed, C.c_void_p(stream)), "ofx_synth_traces")
            return
        _lib.check(self._lib.ofx_synth_traces_psd(
            buf.data_ptr(), None, n, int(lo), self.n_samples, self._t.data_ptr(),
            self._namp.data_ptr(), a_lo, a_hi, frac, dmax, seed, C.c_void_p(stream)),
            "ofx_synth_traces_psd")

    __call__ = fill
