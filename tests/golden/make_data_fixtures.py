#!/usr/bin/env python3
"""Generate golden vectors for the DATA side of the hot path from the reference itself.

Runs ONLY in the authoring container (needs /root/reference); the GPU box and the
test-suite only ever read the .npz it writes.  Nothing from the reference is copied:
the reference's NumPy-only helpers are imported in place and called on a seeded
synthetic `datadb`; inputs and outputs are stored as data.

What is pinned (reference file:line):
  * mycode/utility.py:264-305  reshape2second_stacks  (windowing; enc[:, -1] == dec_in[:, 0])
  * mycode/utility.py:359-446  get_data(pick_user=False / True)
  * mycode/utility.py:483-500  get_gt_target_xyz       (mean / population variance, ddof=0)
  * mycode/utility.py:505-517  get_gt_target_xyz_oth
  * mycode/dataIO.py:16-26     clip_xyz
  * mycode/config.py:6-135     cfg knob values
  * mycode/dataIO.py:77-82     xyz2thetaphi (evaluation side)
  * mycode/given_others_gt_mean_var_seq2seq.py:318-323  _reshape_others_data  (restated inline
    below, because that module executes a training script at import time)

Third-party modules that are absent here (easydict, tensorflow, keras, h5py, statsmodels)
are only imported at module top of the reference files; none of the pinned functions uses
them, so empty in-memory placeholders are enough to let the import statement pass.  The LSTM
arithmetic itself lives in Keras/TensorFlow, which cannot be run here: see oracle/README.md
("parity unpinned" for the cell arithmetic).
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data_helpers.npz")


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_placeholders():
    _placeholder("easydict", EasyDict=_AttrDict)
    _placeholder("tensorflow")
    _placeholder("h5py")
    keras = _placeholder("keras")
    keras.layers = _placeholder("keras.layers", Lambda=lambda f: f)
    keras.utils = _placeholder("keras.utils", to_categorical=None)
    keras.backend = _placeholder("keras.backend")
    sm = _placeholder("statsmodels")
    sm.tsa = _placeholder("statsmodels.tsa")
    sm.tsa.vector_ar = _placeholder("statsmodels.tsa.vector_ar")
    sm.tsa.vector_ar.var_model = _placeholder("statsmodels.tsa.vector_ar.var_model", VAR=None)
    mpl = _placeholder("matplotlib")
    mpl.pyplot = _placeholder("matplotlib.pyplot")


def synthetic_datadb(rng, n_video=2, n_user=3, n_frame=917):
    """Seeded unit-sphere trajectories, slightly over-range so clip_xyz has work to do."""
    db = {}
    for v in range(n_video):
        yaw = rng.uniform(-np.pi, np.pi, (n_user, 1)) + np.cumsum(rng.normal(0, 0.02, (n_user, n_frame)), axis=1)
        pitch = np.clip(rng.normal(0, 0.3, (n_user, 1)) + np.cumsum(rng.normal(0, 0.01, (n_user, n_frame)), axis=1),
                        -np.pi / 2, np.pi / 2)
        scale = 1.0 + 0.02 * rng.standard_normal((n_user, n_frame))
        # values rounded to fp32 (then widened again) so the fixture compresses well
        db["v%02d" % v] = {
            "x": (np.cos(pitch) * np.cos(yaw) * scale).astype(np.float32).astype(np.float64),
            "y": (np.cos(pitch) * np.sin(yaw) * scale).astype(np.float32).astype(np.float64),
            "z": (np.sin(pitch) * scale).astype(np.float32).astype(np.float64),
        }
    return db


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; fixtures can only be regenerated in the authoring container")
    install_placeholders()
    sys.path.insert(0, REF)
    import io
    import contextlib

    from mycode.config import cfg
    from mycode import dataIO
    from mycode import utility as util

    rng = np.random.default_rng(1234)
    datadb_raw = synthetic_datadb(rng)
    out = {}
    for k, v in datadb_raw.items():
        for ax in "xyz":
            out["raw_%s_%s" % (k, ax)] = v[ax].copy()
    datadb = dataIO.clip_xyz({k: {a: v[a].copy() for a in "xyz"} for k, v in datadb_raw.items()})
    for k, v in datadb.items():
        for ax in "xyz":
            out["clip_%s_%s" % (k, ax)] = v[ax].copy()

    # cfg knobs the hot path reads
    knobs = ["batch_size", "fps", "running_length", "predict_step", "predict_len", "data_chunk_stride",
             "input_mean_var", "predict_mean_var", "teacher_forcing", "stateful_across_batch", "shuffle_data",
             "use_one_hot", "sample_and_refeed", "conv_kernel_size", "dropout_rate", "dilation_rate",
             "use_saliency", "time_shift", "cut_data_head", "purelly_testing", "add_xyz_sum1",
             "LEARNING_RATE", "lr_epoch_step", "clip_gradient"]
    out["cfg_names"] = np.array(knobs)
    out["cfg_values"] = np.array([float(cfg[k]) for k in knobs], dtype=np.float64)

    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        enc, fut, fut_in = util.get_data(datadb, pick_user=False)
    out["enc"], out["fut"], out["fut_in"] = enc, fut, fut_in
    assert (enc[:, -1] == fut_in[:, 0]).all()

    out["gt_fut"] = util.get_gt_target_xyz(fut)
    out["gt_fut_in"] = util.get_gt_target_xyz(fut_in)
    out["gt_fut_4d"] = util.get_gt_target_xyz(fut.reshape(fut.shape[0], fut.shape[1], 30, 3))

    np.random.seed(7)  # get_data(pick_user=True) pads missing users with np.random.randint
    with contextlib.redirect_stdout(sink):
        tar, tar_fut, tar_fut_in, oth, oth_fut, oth_fut_in = util.get_data(datadb, pick_user=True, num_user=3)
    out["pu_tar"], out["pu_tar_fut"] = tar, tar_fut
    out["pu_oth_fut"] = oth_fut
    assert (tar[:, -1] == tar_fut_in[:, 0]).all() and (oth[:, :, -1] == oth_fut_in[:, :, 0]).all()

    # _reshape_others_data: (U-1,N,T,90) -> (N,T,U-1,30,3)   (given_others...py:318-323)
    oth_fut5 = oth_fut.transpose((1, 2, 0, 3))
    oth_fut5 = oth_fut5.reshape(oth_fut5.shape[0], oth_fut5.shape[1], oth_fut5.shape[2], 30, 3)
    out["gt_oth_fut"] = util.get_gt_target_xyz_oth(oth_fut5)

    # stride-1 windowing of one small video (exercises shift = T//stride != 1)
    one = np.stack([datadb["v00"][a] for a in "xyz"], axis=-1)[:2, :23 * 30]
    one = one.reshape(one.shape[0], one.shape[1] // 30, 90)
    a, b, c = util.reshape2second_stacks(one, collapse_user=True, stride=1, purelly_testing=False)
    out["s1_in"], out["s1_enc"], out["s1_fut"], out["s1_fut_in"] = one, a, b, c
    a, b, c = util.reshape2second_stacks(one, collapse_user=False, stride=5, purelly_testing=False)
    out["s5_enc"], out["s5_fut"], out["s5_fut_in"] = a, b, c

    # slice_layer on an ndarray (Lambda placeholder returns the python closure)
    sl = util.slice_layer(1, 2, 3)(out["gt_oth_fut"])
    out["slice_1_2_3"] = sl

    # evaluation side: dataIO.xyz2thetaphi (dataIO.py:77-82) on seeded unit vectors
    v = rng.standard_normal((64, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[:4] = [[1, 0, 0], [-1, 1e-9, 0], [0, 0, 1], [0, -1, 0]]
    th, ph = dataIO.xyz2thetaphi(v[:, 0], v[:, 1], v[:, 2])
    out["eval_xyz"], out["eval_theta"], out["eval_phi"] = v, th, ph

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith(("raw_", "clip_"))})


if __name__ == "__main__":
    main()
