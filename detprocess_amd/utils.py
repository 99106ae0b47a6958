"""Host helpers that mirror the reference's channel-expression and window logic.

* ``split_channel_name``   <- detprocess/utils/utils.py:70-184
* ``get_window_indices``   <- detprocess/process/features.py:1243-1344
                              (twin: detprocess/utils/utils.py:189-301)
* ``convert_length_msec_to_samples`` <- pytesio helper used by config.py:471-472
"""

ALLOWED_SEPARATORS = [",", "|", "+", "-"]


def unique_list(items):
    seen = []
    for x in items:
        if x not in seen:
            seen.append(x)
    return seen


def convert_length_msec_to_samples(length_msec, sample_rate):
    return int(round(float(length_msec) * 1e-3 * float(sample_rate)))


def split_channel_name(channel_name, available_channels=None, separator=None, label=None):
    """Split a channel expression; returns (list of channels, separator or None).

    'A' -> (['A'], None); 'A,B' / 'A|B' / 'A+B' -> split on that separator;
    'A-B' needs ``available_channels`` (names may contain '-').
    """
    channel_name = channel_name.replace(" ", "")
    if separator is not None and separator not in ALLOWED_SEPARATORS:
        raise ValueError(f'ERROR: separator "{separator}" not recognized. '
                         f'Allowed separator {ALLOWED_SEPARATORS} ')
    if not any(sep in channel_name for sep in ALLOWED_SEPARATORS):
        return [channel_name], None

    if available_channels is None:
        if separator is None:
            raise ValueError('ERROR: separator required when '
                             '"available_channels" not provided! ')
        if separator == "-":
            raise ValueError('ERROR: "available_channels" required '
                             'when using separator "-"')
        return channel_name.split(separator), separator

    if channel_name in available_channels or channel_name == "all":
        return [channel_name], None

    # which known channels appear, and what is left over must be separators only
    rest = channel_name
    found = []
    for chan in sorted(available_channels, key=len, reverse=True):
        if chan in rest:
            rest = rest.replace(chan, "")
            found.append(chan)
    found.sort(key=channel_name.index)
    leftovers = set(rest)
    bad = [c for c in leftovers if c not in ALLOWED_SEPARATORS]
    if bad:
        raise ValueError(f'ERROR: Unidentified channel "{channel_name}" in yaml file! '
                         f'Perhaps not in raw data? Available channels = '
                         f'{available_channels}')
    seps = sorted(leftovers)
    if separator is None:
        if len(seps) == 1:
            sep = seps[0]
            if sep != "-":
                found = channel_name.split(sep)
            return found, sep
        return found, seps
    if separator not in channel_name:
        return [channel_name], None
    if separator != "-":
        return channel_name.split(separator), separator
    if any(s in channel_name for s in ("|", "+", ",")):
        raise ValueError('Multiple separators available, split first with other '
                         'separators before "-"')
    return found, separator


def get_window_indices(nb_samples, nb_pretrigger_samples, fs,
                       window_min_from_start_usec=None, window_min_to_end_usec=None,
                       window_min_from_trig_usec=None, window_max_from_start_usec=None,
                       window_max_to_end_usec=None, window_max_from_trig_usec=None,
                       **kwargs):
    """us-window -> (min_index, max_index).  int() truncates toward zero BEFORE
    the pretrigger is added; both ends are clamped to [0, nb_samples-1];
    defaults 0 and nb_samples-1; raises if max < min."""
    lo = 0
    if window_min_from_start_usec is not None:
        lo = int(window_min_from_start_usec * fs * 1e-6)
    elif window_min_to_end_usec is not None:
        lo = nb_samples - abs(int(window_min_to_end_usec * fs * 1e-6)) - 1
    elif window_min_from_trig_usec is not None:
        lo = nb_pretrigger_samples + int(window_min_from_trig_usec * fs * 1e-6)
    lo = min(max(lo, 0), nb_samples - 1)
    hi = nb_samples - 1
    if window_max_from_start_usec is not None:
        hi = int(window_max_from_start_usec * fs * 1e-6)
    elif window_max_to_end_usec is not None:
        hi = nb_samples - abs(int(window_max_to_end_usec * fs * 1e-6)) - 1
    elif window_max_from_trig_usec is not None:
        hi = nb_pretrigger_samples + int(window_max_from_trig_usec * fs * 1e-6)
    hi = min(max(hi, 0), nb_samples - 1)
    if hi < lo:
        raise ValueError("ERROR window calculation: max index smaller than min!"
                         "Check configuration!")
    return lo, hi


extract_window_indices = get_window_indices


def cleanup_freq_ranges(f_lims):
    """Normalise ``f_lims`` (psd_amp) into [[f_low, f_high], ...] + range names
    '<low>_<high>' (rounded) -- detprocess/utils/utils.py:437-470."""
    if not isinstance(f_lims, list):
        f_lims = [f_lims]
    ranges, names = [], []
    for fr in f_lims:
        if isinstance(fr, (int, float)):
            fr = [fr]
        f_low = abs(fr[0])
        if len(fr) == 2:
            f_high = abs(fr[1])
            if f_low > f_high:
                f_low, f_high = f_high, f_low
            name = f"{round(f_low)}_{round(f_high)}"
            rng = [f_low, f_high]
        else:
            name = f"{round(f_low)}"
            rng = [f_low]
        if name not in names:
            ranges.append(rng)
            names.append(name)
    return ranges, names


def get_bin_ranges(freq_ranges, nb_samples, fs):
    """One-sided FFT bin ranges [k_lo, k_hi) of psd_amp: the reference indexes the
    DC-dropped folded spectrum (detprocess/utils/utils.py:475-504 applied to
    freqs_fold[1:], algorithms.py:1018-1024), so index i is bin k = i + 1."""
    import numpy as np
    freqs = np.fft.rfftfreq(nb_samples, d=1.0 / fs)[1:]
    out = []
    for fr in freq_ranges:
        lo = int(np.argmin(np.abs(freqs - abs(fr[0]))))
        hi = lo + 1
        if len(fr) == 2:
            hi = int(np.argmin(np.abs(freqs - abs(fr[1]))))
        if lo > hi:
            lo, hi = hi, lo
        if lo == hi:
            if hi < len(freqs) - 1:
                hi += 1
            elif lo > 0:
                lo -= 1
            else:
                raise ValueError("Frequency range too narrow or outside bounds.")
        out.append((lo + 1, hi + 1))
    return out
