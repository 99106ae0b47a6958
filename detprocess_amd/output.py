"""Feature-file writer with the reference's directory and dump naming.

The reference dumps the accumulated rows as ``<group>/<prefix>_<series>_F####.hdf5`` through
``vaex.from_pandas(...).export_hdf5`` whenever the memory limit is hit and at the end
(features.py:584-629; group directory ``<prefix>_I<facility>_D<yyyymmdd>_T<hhmmss>`` and prefix
rules features.py:1030-1082, 494-510).  vaex is not part of this engine; the same columns are
written as an Arrow IPC file (``.arrow``, opened by ``vaex.open`` as a memory-mapped table
exactly like its HDF5 files), as Parquet, or -- when h5py is installed -- as ``.hdf5`` in the
column layout of vaex's own files (``/table/columns/<name>/data``), so that tools globbing for
``*_F####.hdf5`` find the dumps.
"""

import os
import stat
from datetime import datetime

FORMATS = ("arrow", "parquet", "hdf5")


def feature_prefix(processing_id=None, restricted=False, calib=False):
    """'feature', '<id>_feature', with '_restricted' / '_calib' (features.py:494-500)."""
    prefix = "feature" if processing_id is None else f"{processing_id}_feature"
    if restricted:
        prefix += "_restricted"
    elif calib:
        prefix += "_calib"
    return prefix


def series_name_now(facility, now=None):
    """I<facility>_D<yyyymmdd>_T<hhmmss> (features.py:1055-1059)."""
    now = now or datetime.now()
    return f"I{facility}_D{now:%Y%m%d}_T{now:%H%M%S}"


def create_output_directory(base_path, facility, processing_id=None, restricted=False,
                            calib=False, now=None):
    """features.py:1030-1082: ``<base>/<prefix>_<series>`` with group-writable permissions.
    Returns (directory, series name)."""
    series = series_name_now(facility, now)
    out = os.path.join(base_path, f"{feature_prefix(processing_id, restricted, calib)}_{series}")
    if not os.path.isdir(out):
        try:
            os.makedirs(out)
            os.chmod(out, stat.S_IRWXG | stat.S_IRWXU | stat.S_IROTH | stat.S_IXOTH)
        except OSError:
            raise ValueError(f'\nERROR: Unable to create directory "{out}"!\n')
    return out, series


class FeatureWriter:
    """Numbered dumps ``<output_base_file>_F0001.<fmt>``, ``_F0002`` ... (features.py:594-620)."""

    def __init__(self, output_group_path, series_name, processing_id=None, restricted=False,
                 calib=False, fmt="arrow"):
        if fmt not in FORMATS:
            raise ValueError(f'ERROR: output format should be one of {FORMATS}')
        self._fmt = fmt
        self._base = os.path.join(output_group_path,
                                  f"{feature_prefix(processing_id, restricted, calib)}"
                                  f"_{series_name}")
        self._dump = 1
        self.files = []

    def write(self, feature_df):
        """Write one dump (a pandas DataFrame or a dict of equal-length columns)."""
        import pyarrow as pa
        if isinstance(feature_df, dict):
            table = pa.table(feature_df)
        else:
            table = pa.Table.from_pandas(feature_df.reset_index(drop=True),
                                         preserve_index=False)
        name = f"{self._base}_F{str(self._dump).zfill(4)}.{self._fmt}"
        if self._fmt == "hdf5":
            _write_vaex_hdf5(name, table)
        elif self._fmt == "arrow":
            with pa.OSFile(name, "wb") as sink, pa.ipc.new_file(sink, table.schema) as writer:
                writer.write_table(table)
        else:
            import pyarrow.parquet as pq
            pq.write_table(table, name)
        self._dump += 1
        self.files.append(name)
        return name


def _h5py():
    try:
        import h5py
    except ImportError as exc:
        raise ImportError("ERROR: the 'hdf5' feature-file format needs h5py, which is not "
                          "installed here; use fmt='arrow' or fmt='parquet'") from exc
    return h5py


def _write_vaex_hdf5(name, table):
    """The layout ``vaex.DataFrame.export_hdf5`` produces for numeric and string columns
    (features.py:612-616): /table (attr type='table') / columns / <column> / data; strings as
    Arrow-style ``data`` (bytes) + ``indices`` (int64 offsets) with dtype attributes."""
    import numpy as np
    h5py = _h5py()
    with h5py.File(name, "w") as f:
        tab = f.require_group("table")
        tab.attrs["type"] = "table"
        cols = tab.require_group("columns")
        for i, col in enumerate(table.column_names):
            arr = table.column(col).to_numpy(zero_copy_only=False)
            g = cols.require_group(col)
            g.attrs["_order"] = i
            if arr.dtype.kind in "OUS":
                raw = [str(x).encode("utf8") for x in arr]
                off = np.zeros(len(raw) + 1, dtype=np.int64)
                np.cumsum([len(x) for x in raw], out=off[1:])
                g.attrs["type"] = "column"
                d = g.create_dataset("data", data=np.frombuffer(b"".join(raw), dtype=np.uint8))
                d.attrs["dtype"] = "str"
                d.attrs["dtype_item"] = "utf8"
                g.create_dataset("indices", data=off)
            else:
                g.attrs["type"] = "column"
                g.create_dataset("data", data=arr)


def read_features(file_name):
    """Read one dump back as a pandas DataFrame."""
    import pyarrow as pa
    if file_name.endswith(".hdf5"):
        import numpy as np
        import pandas as pd
        h5py = _h5py()
        out = {}
        with h5py.File(file_name, "r") as f:
            cols = f["table/columns"]
            for col in sorted(cols, key=lambda c: cols[c].attrs.get("_order", 0)):
                g = cols[col]
                if "indices" in g:
                    raw, off = bytes(np.asarray(g["data"])), np.asarray(g["indices"])
                    out[col] = [raw[off[i]:off[i + 1]].decode("utf8") for i in range(len(off) - 1)]
                else:
                    out[col] = np.asarray(g["data"])
        return pd.DataFrame(out)
    if file_name.endswith(".parquet"):
        import pyarrow.parquet as pq
        return pq.read_table(file_name).to_pandas()
    with pa.memory_map(file_name, "r") as src:
        return pa.ipc.open_file(src).read_all().to_pandas()
