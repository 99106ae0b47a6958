"""Feature-file writer: the reference's naming (features.py:494-510, 594-620, 1030-1082)."""
import os
from datetime import datetime

import numpy as np
import pandas as pd
import pytest

from detprocess_amd.output import (FeatureWriter, create_output_directory, feature_prefix,
                                   read_features, series_name_now)


def test_names_follow_the_reference():
    assert feature_prefix() == "feature"
    assert feature_prefix("run28", restricted=True) == "run28_feature_restricted"
    assert feature_prefix(None, calib=True) == "feature_calib"
    assert feature_prefix("x", restricted=True, calib=True) == "x_feature_restricted"
    now = datetime(2023, 6, 30, 19, 37, 42)
    assert series_name_now(2, now) == "I2_D20230630_T193742"


@pytest.mark.parametrize("fmt", ["arrow", "parquet"])
def test_numbered_dumps_round_trip(tmp_path, fmt):
    out, series = create_output_directory(str(tmp_path), 2, processing_id="proc",
                                          now=datetime(2023, 6, 30, 19, 37, 42))
    assert os.path.basename(out) == "proc_feature_I2_D20230630_T193742" and os.path.isdir(out)
    w = FeatureWriter(out, series, processing_id="proc", fmt=fmt)
    rng = np.random.default_rng(0)
    df1 = pd.DataFrame({"amp_of1x1_nodelay_chanA": rng.normal(size=5),
                        "event_number": np.arange(5, dtype=np.int64)})
    df2 = {"amp_of1x1_nodelay_chanA": np.full(3, -999999.0), "event_number": np.arange(5, 8)}
    f1, f2 = w.write(df1), w.write(df2)
    assert os.path.basename(f1) == f"proc_feature_I2_D20230630_T193742_F0001.{fmt}"
    assert os.path.basename(f2) == f"proc_feature_I2_D20230630_T193742_F0002.{fmt}"
    assert w.files == [f1, f2]
    back = read_features(f1)
    assert list(back.columns) == list(df1.columns)
    assert np.array_equal(back["amp_of1x1_nodelay_chanA"].to_numpy(),
                          df1["amp_of1x1_nodelay_chanA"].to_numpy())
    assert np.array_equal(read_features(f2)["event_number"].to_numpy(), np.arange(5, 8))
    with pytest.raises(ValueError):
        FeatureWriter(out, series, fmt="csv")


def test_hdf5_formats_need_h5py_and_round_trip_when_it_is_there(tmp_path):
    """The reference's containers (pytesio FilterH5IO files, vaex export_hdf5 dumps) need h5py,
    which this image lacks: the loaders say so instead of failing obscurely; where h5py exists
    the files round-trip."""
    import numpy as np
    import pytest
    from detprocess_amd import FilterData, synth
    from detprocess_amd.output import FeatureWriter, read_features
    n, fs = 512, 1.25e6
    fd = FilterData()
    fd.set_template("A", synth.make_template(n, n // 2, fs), sample_rate=fs,
                    pretrigger_length_samples=n // 2, tag="default")
    fd.set_psd("A", synth.make_psd(n, fs), np.fft.fftfreq(n, d=1 / fs), sample_rate=fs)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py is required"):
            fd.save_hdf5(str(tmp_path / "f.hdf5"))
        with pytest.raises(ImportError, match="h5py is required"):
            FilterData().load_hdf5(str(tmp_path / "f.hdf5"))
        w = FeatureWriter(str(tmp_path), "I2_D20240101_T000000", fmt="hdf5")
        with pytest.raises(ImportError, match="needs h5py"):
            w.write({"a": np.arange(3.0)})
        return
    fd.save_hdf5(str(tmp_path / "f.hdf5"))
    fd2 = FilterData()
    fd2.load_hdf5(str(tmp_path / "f.hdf5"))
    t, _, m = fd2.get_template("A", return_metadata=True)
    assert np.array_equal(t, fd.get_template("A")[0]) and m["nb_pretrigger_samples"] == n // 2
    p2, f2 = fd2.get_psd("A")
    assert np.array_equal(p2, fd.get_psd("A")[0]) and np.array_equal(f2, fd.get_psd("A")[1])
    w = FeatureWriter(str(tmp_path), "I2_D20240101_T000000", fmt="hdf5")
    name = w.write({"amp": np.arange(4.0), "chan": ["a", "bb", "", "d"]})
    assert name.endswith("_F0001.hdf5")
    df = read_features(name)
    assert list(df["amp"]) == [0.0, 1.0, 2.0, 3.0] and list(df["chan"]) == ["a", "bb", "", "d"]
