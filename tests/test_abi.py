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


_C2CT = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "int64_t": ctypes.c_int64, "long long": ctypes.c_int64,
         "float": ctypes.c_float, "double": ctypes.c_double, "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint,
         "uint64_t": ctypes.c_uint64, "unsigned long long": ctypes.c_uint64, "fov_stream_t": ctypes.c_void_p}


def _header_prototypes():
    """name -> (return ctype, [parameter ctypes]) parsed from the header's declarations: a pointer of any kind is c_void_p
    (const char* as a RETURN type is c_char_p), scalars map by their C type."""
    text = open(os.path.join(ROOT, "include", "fov360.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b(fov_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret = ret.strip()
        rt = ctypes.c_char_p if ret.replace(" ", "") == "constchar*" else (ctypes.c_void_p if "*" in ret else
                                                                           (None if ret == "void" else _C2CT[ret]))
        args = []
        for p in (q.strip() for q in params.split(",")):
            if p in ("", "void"):
                continue
            if "*" in p:
                args.append(ctypes.c_void_p)
                continue
            words = [w for w in re.sub(r"\bconst\b", "", p).split()]
            ctype = " ".join(words[:-1]) if len(words) > 1 else words[0]      # drop the parameter name
            assert ctype in _C2CT, "fov360.h: %s has a parameter of unmapped type %r" % (name, p)
            args.append(_C2CT[ctype])
        protos[name] = (rt, args)
    return protos


def test_ctypes_argtypes_match_header_parameter_lists():
    """Every _lib.SIGNATURES entry against the header's declaration: the same NUMBER of parameters and, one by one, the same
    ctypes class (pointer / int / size_t / int64 / float / double) and return type - a table that drifts from the header's
    parameter list (an argument added on one side only) fails here, not as a mis-read register on the GPU."""
    protos = _header_prototypes()
    assert sorted(protos) == _declared()
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        h_ret, h_args = protos[name]
        assert len(argtypes) == len(h_args), "%s: ctypes table has %d arguments, the header %d" % (name, len(argtypes), len(h_args))
        for i, (a, b) in enumerate(zip(argtypes, h_args)):
            if b is ctypes.c_void_p and issubclass(a, ctypes._Pointer):     # POINTER(c_void_p) for an out-parameter: a pointer
                continue
            assert a is b, "%s: argument %d is %s in the ctypes table, %s in the header" % (name, i, a.__name__, b.__name__)
        assert restype is h_ret, "%s: return type %s vs header %s" % (name, restype, h_ret)


def test_deferral_registry_is_keyed_and_bounded_without_a_gpu():
    """fov_reduce_defer_begin / _flush / _end keep host-side bookkeeping only until a product records something: regions are keyed
    by their gradient buffer, 16 may be open, the 17th is refused with a message, _end(NULL) closes all (no launch happens with no
    pending record, so this runs on a box without a GPU)."""
    L = _lib.lib()
    base = 0x7f0000000000
    arena = ctypes.c_void_p(base + (1 << 30))
    for i in range(16):
        assert L.fov_reduce_defer_begin(ctypes.c_void_p(base + i * 4096), 1024, arena, 1 << 20, None) == 0
    assert L.fov_reduce_defer_begin(ctypes.c_void_p(base + 16 * 4096), 1024, arena, 1 << 20, None) == _lib.ERR_UNSUPPORTED
    assert b"too many open regions" in L.fov_last_error()
    assert L.fov_reduce_defer_flush(ctypes.c_void_p(base + 3 * 4096 + 8), None) == 0       # any address inside a region names it
    assert L.fov_reduce_defer_end(ctypes.c_void_p(base + 3 * 4096), None) == 0
    assert L.fov_reduce_defer_begin(ctypes.c_void_p(base + 16 * 4096), 1024, arena, 1 << 20, None) == 0   # a slot is free again
    # a region over part of an open one replaces it (still 16 open: the next begin on fresh memory is refused)
    assert L.fov_reduce_defer_begin(ctypes.c_void_p(base + 5 * 4096 + 64), 16, arena, 1 << 20, None) == 0
    assert L.fov_reduce_defer_begin(ctypes.c_void_p(base + 40 * 4096), 1024, arena, 1 << 20, None) == _lib.ERR_UNSUPPORTED
    assert L.fov_reduce_defer_end(None, None) == 0
    assert L.fov_reduce_defer_begin(ctypes.c_void_p(base + 40 * 4096), 1024, arena, 1 << 20, None) == 0
    assert L.fov_reduce_defer_end(None, None) == 0
    assert L.fov_reduce_defer_begin(ctypes.c_void_p(base), 1024, None, 0, None) == _lib.ERR_INVALID     # a buffer without an arena
