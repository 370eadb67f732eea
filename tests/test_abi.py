"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/fov360.h declares; the ctypes table mirrors the header."""
import ctypes
import os
import re

from longterm360fov_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "fov360.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fov_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    names = _declared()
    assert "fov_seq2seq_decode_fwd" in names and "fov_lstm_seq_fwd" in names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "libfov360_hip.so does not export " + n


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_no_compute_queries_without_gpu():
    L = _lib.lib()
    assert L.fov_version() >= 100
    assert L.fov_cluster_supported(90, 256) == 1
    assert L.fov_cluster_supported(90, 48) == 0
    assert L.fov_cluster_supported(200, 256) == 0
    assert L.fov_last_error() is not None


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under longterm360fov_amd/ may reference it."""
    pkg = os.path.join(ROOT, "longterm360fov_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
