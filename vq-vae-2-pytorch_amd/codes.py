"""On-disk format of the extracted codes: what /root/reference/extract_code.py:27-33 writes and
dataset.py:25-51 (`LMDBDataset`) reads back.

  row i   : key str(i).encode()  ->  pickle.dumps(CodeRow(top=id_t[i], bottom=id_b[i], filename=name_i))
            with CodeRow = namedtuple('CodeRow', ['top', 'bottom', 'filename']) living in module `dataset`
            (dataset.py:11) and top / bottom int64 numpy arrays
  length  : key b'length'        ->  str(number of rows).encode()

The ROW BYTES are the reference's, bit for bit (tests/test_host_cpu.py builds the expected pickle from the field
spec alone).  The container is LMDB in the reference; `lmdb` is not importable in this environment, so CodeStore
writes the same key -> bytes pairs into LMDB when the module exists and into a single-file sqlite3 table otherwise
(same keys, same values; a maintainer with lmdb installed gets the reference's container unchanged).
"""
import collections
import contextlib
import os
import pickle
import sqlite3
import sys
import types

import numpy as np
import torch

_FIELDS = ["top", "bottom", "filename"]   # dataset.py:11


@contextlib.contextmanager
def _code_row_class():
    """The class the reference's readers unpickle: `dataset.CodeRow`.  If the reference's dataset.py is importable
    use it; otherwise lend the name to an identical namedtuple for the duration of the dump (pickle stores only the
    module/qualname reference, so the bytes are the same either way)."""
    mod = sys.modules.get("dataset")
    if mod is not None and hasattr(mod, "CodeRow"):
        yield mod.CodeRow
        return
    stub = types.ModuleType("dataset")
    row = collections.namedtuple("CodeRow", _FIELDS)
    row.__module__ = "dataset"
    stub.CodeRow = row
    had = sys.modules.get("dataset")
    sys.modules["dataset"] = stub
    try:
        yield row
    finally:
        if had is None:
            del sys.modules["dataset"]
        else:
            sys.modules["dataset"] = had


def code_row_bytes(top, bottom, filename):
    """pickle.dumps(CodeRow(top, bottom, filename)) exactly as extract_code.py:28-29 produces it."""
    top = np.asarray(top.cpu() if isinstance(top, torch.Tensor) else top)
    bottom = np.asarray(bottom.cpu() if isinstance(bottom, torch.Tensor) else bottom)
    with _code_row_class() as row:
        return pickle.dumps(row(top=top, bottom=bottom, filename=filename))


_ROW = collections.namedtuple("CodeRow", _FIELDS)

# the only globals a row pickle may reference: the row type itself and what numpy needs to rebuild an ndarray
_ALLOWED_GLOBALS = {("dataset", "CodeRow")} | {
    (mod, name) for mod in ("numpy.core.multiarray", "numpy._core.multiarray") for name in ("_reconstruct",)
} | {("numpy", "ndarray"), ("numpy", "dtype")}


class _RowUnpickler(pickle.Unpickler):
    """A code store may come from elsewhere (including one written by other tooling): rows hold a namedtuple of two
    int64 ndarrays and a str, so nothing else is allowed to resolve -- a blob that names any other global raises
    instead of importing / calling it."""

    def find_class(self, module, name):
        if (module, name) not in _ALLOWED_GLOBALS:
            raise pickle.UnpicklingError(f"code row refers to {module}.{name}: not a CodeRow of numpy arrays")
        if (module, name) == ("dataset", "CodeRow"):
            return _ROW
        return super().find_class(module, name)


def load_code_row(blob):
    """(top, bottom, filename) from a row written by the reference or by code_row_bytes (restricted unpickling)."""
    import io
    r = _RowUnpickler(io.BytesIO(blob)).load()
    if not (isinstance(r, _ROW) and isinstance(r.top, np.ndarray) and isinstance(r.bottom, np.ndarray)
            and isinstance(r.filename, str) and r.top.dtype != object and r.bottom.dtype != object):
        raise pickle.UnpicklingError("code row is not CodeRow(top ndarray, bottom ndarray, filename str)")
    return r.top, r.bottom, r.filename


class CodeStore:
    """Key -> bytes container with the put/get surface extract_code.py uses on an LMDB transaction."""

    def __init__(self, path, mode="r", backend="auto", map_size=100 * 1024 ** 3):
        if backend == "auto":
            if mode == "r" and os.path.isdir(path):
                backend = "lmdb"            # an LMDB environment is a directory
            elif mode == "r":
                backend = "sqlite"
            else:
                try:
                    import lmdb  # noqa: F401
                    backend = "lmdb"
                except ImportError:
                    backend = "sqlite"
        self.backend, self.mode = backend, mode
        if backend == "lmdb":
            import lmdb
            if mode == "w":
                self.env = lmdb.open(path, map_size=map_size)                      # extract_code.py:66-68
            else:
                self.env = lmdb.open(path, max_readers=32, readonly=True, lock=False, readahead=False,
                                     meminit=False)                                 # dataset.py:27-34
            self.txn = self.env.begin(write=(mode == "w"))
        elif backend == "sqlite":
            if mode == "r" and not os.path.exists(path):
                raise IOError("Cannot open code store", path)                       # dataset.py:36-37
            self.db = sqlite3.connect(path)
            self.db.execute("CREATE TABLE IF NOT EXISTS kv (k BLOB PRIMARY KEY, v BLOB NOT NULL)")
        else:
            raise ValueError(f"unknown backend {backend!r}")

    def put(self, key, value):
        if self.mode != "w":
            raise IOError("code store opened read-only")
        if self.backend == "lmdb":
            self.txn.put(key, value)
        else:
            self.db.execute("INSERT OR REPLACE INTO kv VALUES (?, ?)", (key, value))

    def get(self, key):
        if self.backend == "lmdb":
            return self.txn.get(key)
        row = self.db.execute("SELECT v FROM kv WHERE k = ?", (key,)).fetchone()
        return None if row is None else bytes(row[0])

    def close(self):
        if self.backend == "lmdb":
            if self.mode == "w":
                self.txn.commit()
            self.env.close()
        else:
            self.db.commit()
            self.db.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def write_code_rows(store, tops, bottoms, filenames, start=0):
    """One row per image (extract_code.py:27-31); returns the next free index."""
    index = start
    for top, bottom, name in zip(tops, bottoms, filenames):
        store.put(str(index).encode("utf-8"), code_row_bytes(top, bottom, name))
        index += 1
    return index


class CodeDataset(torch.utils.data.Dataset):
    """dataset.py:25-51 (LMDBDataset) over a CodeStore: item -> (top tensor, bottom tensor, filename)."""

    def __init__(self, path, backend="auto"):
        self.store = CodeStore(path, "r", backend)
        length = self.store.get("length".encode("utf-8"))
        if length is None:
            raise IOError("Cannot open code store", path)
        self.length = int(length.decode("utf-8"))

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        top, bottom, filename = load_code_row(self.store.get(str(index).encode("utf-8")))
        return torch.from_numpy(top), torch.from_numpy(bottom), filename
