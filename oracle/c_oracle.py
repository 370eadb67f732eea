"""ctypes loader for the C oracle (oracle/lstm_ref.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def lstm_layer(x, K, R, b, h0=None, c0=None, act=0):
    x, K, R, b, h0, c0 = map(_f32, (x, K, R, b, h0, c0))
    B, T, F = x.shape
    H = R.shape[0]
    hs = np.empty((B, T, H), np.float32)
    hT = np.empty((B, H), np.float32)
    cT = np.empty((B, H), np.float32)
    rc = lib().oracle_lstm_layer_f32(_p(x), _p(K), _p(R), _p(b), _p(h0), _p(c0), _p(hs), _p(hT), _p(cT),
                                     B, T, F, H, int(act))
    assert rc == 0, rc
    return hs, hT, cT


def _w(w):
    return [_f32(w[k]) for k in ("enc_K", "enc_R", "enc_b", "dec_K", "dec_R", "dec_b", "dense_W", "dense_b")]


def seq2seq_decode(enc_in, dec_in0, w, T_out, act=0):
    enc_in, dec_in0 = _f32(enc_in), _f32(dec_in0)
    ws = _w(w)
    B, T_in, F_enc = enc_in.shape
    F_dec = ws[6].shape[1]
    H = ws[1].shape[0]
    out = np.empty((B, T_out, F_dec), np.float32)
    rc = lib().oracle_seq2seq_decode_f32(_p(enc_in), _p(dec_in0), *[_p(a) for a in ws], _p(out),
                                         B, T_in, T_out, F_enc, F_dec, H, int(act))
    assert rc == 0, rc
    return out


def seq2seq_teacher_forced(enc_in, dec_in, w, act=0):
    enc_in, dec_in = _f32(enc_in), _f32(dec_in)
    ws = _w(w)
    B, T_in, F_enc = enc_in.shape
    T_out, F_dec = dec_in.shape[1], dec_in.shape[2]
    H = ws[1].shape[0]
    out = np.empty((B, T_out, F_dec), np.float32)
    rc = lib().oracle_seq2seq_tf_f32(_p(enc_in), _p(dec_in), *[_p(a) for a in ws], _p(out),
                                     B, T_in, T_out, F_enc, F_dec, H, int(act))
    assert rc == 0, rc
    return out
