"""ctypes binding of libofx.so (the C ABI in include/ofx.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, this module raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C detprocess_amd/csrc``.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFX_LIB", os.path.join(_HERE, "libofx.so"))

OK = 0
MEM_HOST, MEM_DEVICE = 0, 1
ENGINE_AUTO, ENGINE_FUSED, ENGINE_ROCFFT, ENGINE_LDS = 0, 1, 2, 3
SEARCH_NODELAY, SEARCH_DELAY, SEARCH_DELAY_INTERP = 0, 1, 2
SEARCH_FLOATS, TDWIN_FLOATS, BAND_FLOATS = 8, 8, 1
MAX_SLOTS, MAX_SEARCHES, MAX_TDWIN, MAX_TERMS, MAX_BANDS = 8, 8, 8, 8, 16
COL = {"amp": 0, "t0": 1, "chi2": 2, "lowchi2": 3, "chi2nopulse": 4,
       "ampres": 5, "timeres": 6, "index": 7}
TD = {"baseline": 0, "integral": 1, "maximum": 2, "minimum": 3, "sum": 4, "sumsq": 5,
      "first": 6, "last": 7}

# every symbol include/ofx.h declares: (name, restype, argtypes)
_p = C.c_void_p
_SYMBOLS = [
    ("ofx_last_error", C.c_char_p, []),
    ("ofx_device_info", C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    ("ofx_plan_create", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_double,
                                  C.c_int, C.c_int, C.c_int]),
    ("ofx_plan_destroy", C.c_int, [_p]),
    ("ofx_plan_engine", C.c_int, [_p]),
    ("ofx_plan_set_filter", C.c_int, [_p, C.c_int, _p, _p, _p, C.c_double, C.c_double]),
    ("ofx_plan_add_search", C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_double]),
    ("ofx_plan_add_tdwindow", C.c_int, [_p, C.c_int, C.c_int]),
    ("ofx_plan_add_band", C.c_int, [_p, C.c_int, C.c_int]),
    ("ofx_plan_set_channels", C.c_int, [_p, C.c_int, C.c_int, _p, _p]),
    ("ofx_plan_reset", C.c_int, [_p]),
    ("ofx_plan_row_floats", C.c_int, [_p]),
    ("ofx_plan_search_offset", C.c_int, [_p, C.c_int, C.c_int]),
    ("ofx_plan_tdwindow_offset", C.c_int, [_p, C.c_int]),
    ("ofx_plan_band_offset", C.c_int, [_p, C.c_int]),
    ("ofx_process", C.c_int, [_p, _p, _p, C.c_longlong, C.c_int, _p, C.c_int, _p]),
    ("ofx_process_adc", C.c_int, [_p, _p, C.c_longlong, C.c_int, _p, C.c_longlong, _p, _p, _p,
                                  C.c_int, _p]),
    ("ofx_trigger_create", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_double, _p, C.c_double,
                                     C.c_double, C.c_int]),
    ("ofx_trigger_create_nxm", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_double, C.c_int,
                                         C.c_int, _p, _p, _p, C.c_int]),
    ("ofx_trigger_destroy", C.c_int, [_p]),
    ("ofx_trigger_update_traces", C.c_int, [_p, _p, C.c_int, C.c_longlong, C.c_int, _p, _p,
                                            C.c_int, _p]),
    ("ofx_trigger_update_trace", C.c_int, [_p, _p, C.c_int, C.c_longlong, C.c_int, C.c_double,
                                           C.c_double, C.c_int, _p]),
    ("ofx_trigger_get_traces", C.c_int, [_p, _p, _p, C.c_int, _p]),
    ("ofx_trigger_find", C.c_int, [_p, C.c_double, C.c_longlong, _p, _p, _p, C.c_longlong,
                                   C.POINTER(C.c_longlong), _p]),
    ("ofx_trigger_above", C.c_int, [_p, C.c_double, _p, _p, C.c_longlong,
                                    C.POINTER(C.c_longlong), _p]),
    ("ofx_trigger_gather", C.c_int, [_p, _p, C.c_longlong, _p, _p, _p]),
    ("ofx_trigger_set_pulse_table", C.c_int, [_p, _p]),
    ("ofx_trigger_residual_subtract", C.c_int, [_p, _p, C.c_longlong, _p]),
    ("ofx_trigger_residual_restore", C.c_int, [_p, _p, C.c_int, _p]),
    ("ofx_nxm_create", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                 C.c_int, C.c_int]),
    ("ofx_nxm_destroy", C.c_int, [_p]),
    ("ofx_nxm_set_filter", C.c_int, [_p, _p, _p, _p]),
    ("ofx_nxm_add_search", C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("ofx_nxm_reset_searches", C.c_int, [_p]),
    ("ofx_nxm_set_channels", C.c_int, [_p, C.c_int, _p]),
    ("ofx_nxm_row_floats", C.c_int, [_p]),
    ("ofx_nxm_process", C.c_int, [_p, _p, _p, C.c_longlong, C.c_int, _p, C.c_int, _p]),
    ("ofx_nxm_process_adc", C.c_int, [_p, _p, C.c_longlong, C.c_int, _p, C.c_longlong, _p, _p, _p,
                                      C.c_int, _p]),
    ("ofx_synth_traces", C.c_int, [_p, _p, C.c_longlong, C.c_longlong, C.c_int, _p,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                   C.c_ulonglong, _p]),
    ("ofx_synth_traces_psd", C.c_int, [_p, _p, C.c_longlong, C.c_longlong, C.c_int, _p, _p,
                                       C.c_float, C.c_float, C.c_float, C.c_int,
                                       C.c_ulonglong, _p]),
    ("ofx_synth_release", C.c_int, []),
    ("ofx_plan_kernel_time", C.c_int, [_p, C.POINTER(C.c_double),
                                       C.POINTER(C.c_longlong)]),
    ("ofx_plan_enable_timing", C.c_int, [_p, C.c_int]),
]
SYMBOL_NAMES = [s[0] for s in _SYMBOLS]

_lib = None


class OfxError(RuntimeError):
    pass


def load():
    """Load libofx.so and bind every declared symbol (raises if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OfxError(
            f"ERROR: {LIB_PATH} not found. The HIP extension is required "
            "(no CPU fallback). Build it: make -C detprocess_amd/csrc")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in _SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != OK:
        msg = load().ofx_last_error().decode(errors="replace")
        if rc == 1:
            raise ValueError(f"ERROR: {what}: {msg}")
        raise OfxError(f"ERROR: {what}: rc={rc}: {msg}")
