#!/usr/bin/env python3
"""Writes tests/golden/keras_seq2seq_tiny.h5: a Keras-2.2-layout HDF5 weight file (`model.save_weights`) of a tiny
FoV_seq2seq model (F_enc 5, F_dec 3, latent_dim 4; layers input_1, input_2, lstm_1, lstm_2, dense_1 - the inputs carry
no weights, as in a real file), plus keras_seq2seq_tiny_full.h5 (the `model.save` form: the same tree under
`model_weights/`), with seeded values that tests/test_host.py regenerates.

The bytes come from the package's own writer (longterm360fov_amd/keras_h5.py); the real HDF5 library's h5ls / h5dump
(1.10.6, authoring container) list and dump them correctly.  The reader's ground truth are the files the real library
wrote: make_keras_h5_real.c -> keras_real_weights.h5, keras_real_model.h5.
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from longterm360fov_amd import keras_h5  # noqa: E402


def tiny_layers(seed=7, F_enc=5, F_dec=3, H=4):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return [("input_1", []), ("input_2", []),
            ("lstm_1", [("lstm_1/kernel:0", f(F_enc, 4 * H)), ("lstm_1/recurrent_kernel:0", f(H, 4 * H)), ("lstm_1/bias:0", f(4 * H))]),
            ("lstm_2", [("lstm_2/kernel:0", f(F_dec, 4 * H)), ("lstm_2/recurrent_kernel:0", f(H, 4 * H)), ("lstm_2/bias:0", f(4 * H))]),
            ("dense_1", [("dense_1/kernel:0", f(H, F_dec)), ("dense_1/bias:0", f(F_dec))])]


def wrap_as_full_model(src, dst):
    """model.save() layout: a root group holding `model_weights`.  Built by re-basing: the weights file is embedded
    behind a new superblock whose root group has the single member `model_weights` -> the old root object header."""
    layers = keras_h5.read_keras_layers(src)
    w = keras_h5._Writer()
    w.alloc(96)
    layer_addr = {}
    maxm = 4
    for lname, ws in layers:
        tree = {}
        for wn, arr in ws:
            parts = wn.split("/")
            node = tree
            for prt in parts[:-1]:
                node = node.setdefault(prt, {})
            node[parts[-1]] = keras_h5._write_dataset(w, arr)

        def emit(node, attrs):
            members = {k: (emit(v, {}) if isinstance(v, dict) else v) for k, v in node.items()}
            return keras_h5._write_group(w, members, attrs, 4)[0]

        names = np.array([wn.encode() for wn, _ in ws], dtype="S") if ws else np.zeros((0,), dtype="S1")
        layer_addr[lname] = emit(tree, {"weight_names": names})
        maxm = max(maxm, len(layers))
    mw = keras_h5._write_group(w, layer_addr, {"layer_names": np.array([n.encode() for n, _ in layers], dtype="S"),
                                               "backend": np.array(b"tensorflow", dtype="S"),
                                               "keras_version": np.array(b"2.2.4", dtype="S")}, 4)[0]
    root, btree, heap = keras_h5._write_group(w, {"model_weights": mw}, {"keras_version": np.array(b"2.2.4", dtype="S"),
                                                                        "backend": np.array(b"tensorflow", dtype="S")}, 4)
    sb = keras_h5.SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, keras_h5.UNDEF, len(w.buf), keras_h5.UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
    w.put(0, sb)
    with open(dst, "wb") as fh:
        fh.write(bytes(w.buf))


if __name__ == "__main__":
    a = os.path.join(HERE, "keras_seq2seq_tiny.h5")
    keras_h5.write_keras_layers(a, tiny_layers())
    wrap_as_full_model(a, os.path.join(HERE, "keras_seq2seq_tiny_full.h5"))
    print("wrote", a, os.path.getsize(a), "bytes")
