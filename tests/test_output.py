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
        FeatureWriter(out, series, fmt="hdf5")
