#!/usr/bin/env python3
"""Thin command-line driver for the feature stage (the ``--enable-feature`` leg of the
reference's scripts/process.py:742-789), for data that is already in array form: the raw
pytesdaq HDF5 reader lives in pytesio, which this engine does not replace.

  python scripts/process_features.py --processing_setup process.yaml --filter_file filter.npz \\
         --events events.npy --channels chanA,chanB --sample_rate 1.25e6 --save_path out/

  --events   float32 [n_events, n_channels, n_samples] (.npy, memory-mapped), or with --adc an
             int16 stream file [n_channels, n_stream] plus --trigger_index (.npy int64) and
             --adc_scale / --adc_offset (comma lists, amps = adc * scale + offset)
  --filter_file  FilterData.save_npz file (templates, PSDs / CSDs per channel and tag)

Dumps are written as <save_path>/<prefix>_I<facility>_D..._T.../<prefix>_<series>_F0001.arrow
(detprocess_amd/output.py), one dump per --events_per_dump events.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _floats(text, n):
    vals = [float(v) for v in text.split(",")]
    if len(vals) == 1:
        vals = vals * n
    if len(vals) != n:
        raise SystemExit(f"expected 1 or {n} comma-separated values, got {len(vals)}")
    return np.array(vals)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument("--processing_setup", required=True, help="YAML configuration")
    ap.add_argument("--filter_file", required=True, help="FilterData .npz file")
    ap.add_argument("--events", required=True, help=".npy events, or the int16 streams with --adc")
    ap.add_argument("--channels", required=True, help="comma list: names of axis 1 of the events")
    ap.add_argument("--sample_rate", type=float, required=True)
    ap.add_argument("--save_path", required=True)
    ap.add_argument("--adc", action="store_true", help="--events holds int16 streams")
    ap.add_argument("--trigger_index", help=".npy int64 trigger indices (with --adc)")
    ap.add_argument("--adc_scale", default="1.0")
    ap.add_argument("--adc_offset", default="0.0")
    ap.add_argument("--nb_samples", type=int, help="trace length (with --adc, if not in the YAML)")
    ap.add_argument("--nb_pretrigger_samples", type=int)
    ap.add_argument("--events_per_dump", type=int, default=200000)
    ap.add_argument("--processing_id")
    ap.add_argument("--restricted", action="store_true")
    ap.add_argument("--calib", action="store_true")
    ap.add_argument("--facility", type=int, default=1)
    ap.add_argument("--format", choices=("arrow", "parquet"), default="arrow")
    ap.add_argument("--external_file")
    ap.add_argument("--skip_unsupported", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    from detprocess_amd import FeatureProcessing, FilterData
    from detprocess_amd.output import FeatureWriter, create_output_directory

    channels = [c.strip() for c in args.channels.split(",")]
    fd = FilterData()
    fd.load_npz(args.filter_file)
    fp = FeatureProcessing(args.processing_setup, fd, channels, args.sample_rate,
                           nb_samples=args.nb_samples,
                           nb_pretrigger_samples=args.nb_pretrigger_samples, device=args.device,
                           external_file=args.external_file,
                           skip_unsupported=args.skip_unsupported)
    out_dir, series = create_output_directory(args.save_path, args.facility, args.processing_id,
                                              args.restricted, args.calib)
    writer = FeatureWriter(out_dir, series, args.processing_id, args.restricted, args.calib,
                           fmt=args.format)
    data = np.load(args.events, mmap_mode="r")
    step = max(1, args.events_per_dump)
    total = 0
    if args.adc:
        if not args.trigger_index:
            raise SystemExit("--adc needs --trigger_index")
        trig = np.load(args.trigger_index).astype(np.int64)
        scale = _floats(args.adc_scale, len(channels))
        offset = _floats(args.adc_offset, len(channels))
        adc = np.ascontiguousarray(data)
        for b0 in range(0, len(trig), step):
            df = fp.process_adc(adc, trig[b0:b0 + step], scale, offset, n_samples=args.nb_samples)
            df.insert(0, "event_index", np.arange(b0, b0 + len(df), dtype=np.int64))
            df.insert(1, "trigger_index", trig[b0:b0 + step])
            print("INFO: wrote", writer.write(df), f"({len(df)} events)")
            total += len(df)
    else:
        if data.ndim == 2:
            data = data.reshape(data.shape[0], 1, data.shape[1])
        for b0 in range(0, data.shape[0], step):
            df = fp.process(np.ascontiguousarray(data[b0:b0 + step], dtype=np.float32))
            df.insert(0, "event_index", np.arange(b0, b0 + len(df), dtype=np.int64))
            print("INFO: wrote", writer.write(df), f"({len(df)} events)")
            total += len(df)
    print(f"Processing done! {total} events, {len(writer.files)} file(s) in {out_dir}")
    return writer.files


if __name__ == "__main__":
    main()
