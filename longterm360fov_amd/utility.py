"""Host-side data helpers of the hot path (NumPy; they stay Python in the reference too).

Own restatement of the reference's windowing / feature helpers with identical shapes and
results (pinned by tests/golden/data_helpers.npz, which was produced by running the
reference's functions):
    clip_xyz                 mycode/dataIO.py:16-26
    reshape2second_stacks    mycode/utility.py:264-305
    get_data                 mycode/utility.py:359-446
    get_gt_target_xyz[_oth]  mycode/utility.py:483-517
    slice_layer              mycode/utility.py:246-261
    reshape_others_data      mycode/given_others_gt_mean_var_seq2seq.py:318-323
    generator_train2         mycode/data_generator_including_saliency.py:114-182
"""
import numpy as np

from .config import cfg


def clip_xyz(datadb):
    """Clip every x/y/z track to [-1, 1] in place and return the dict."""
    for vid in datadb.keys():
        for ax in ("x", "y", "z"):
            np.clip(datadb[vid][ax], -1, 1, out=datadb[vid][ax])
    return datadb


def cut_head_or_tail_less_than_1sec(per_video_db, fps=None):
    """Drop the partial second at the head (cfg.cut_data_head) or tail so frames % fps == 0."""
    fps = fps or cfg.fps
    n = per_video_db.shape[1]
    extra = n - (n // fps) * fps
    if extra == 0:
        return per_video_db
    return per_video_db[:, extra:] if cfg.cut_data_head else per_video_db[:, :n - extra]


def reshape2second_stacks(per_video_db, collapse_user=False, stride=None, purelly_testing=None):
    """(U, S, feat) seconds -> windows of cfg.running_length seconds every `stride` seconds.

    Returns (enc, future, future_input); with collapse_user the user and window axes are merged
    window-major -> (M, T, feat), else the user axis leads -> (U, W, T, feat).  The decoder
    input is the future shifted right by one second, seeded with the encoder's last second.
    """
    T = cfg.running_length
    stride = cfg.data_chunk_stride if stride is None else stride
    purelly_testing = cfg.purelly_testing if purelly_testing is None else purelly_testing
    feat = per_video_db.shape[-1]
    assert per_video_db.shape[1] >= 2 * T
    shift = T // stride
    if purelly_testing:
        pad = np.zeros((per_video_db.shape[0], shift, feat))
        per_video_db = np.concatenate((per_video_db, pad), axis=1)
    nrows = (per_video_db.shape[1] - T) // stride + 1
    idx = stride * np.arange(nrows)[:, None] + np.arange(T)           # (W, T)
    win = np.asarray(per_video_db, dtype=np.float64)[:, idx, :]       # (U, W, T, feat)
    win = win.transpose(1, 0, 2, 3)                                   # (W, U, T, feat)
    fut = win[shift:]
    enc = win[:-shift]
    fut_in = np.concatenate((enc[:, :, -1:, :], fut[:, :, :-1, :]), axis=2)
    if collapse_user:
        return (enc.reshape(-1, T, feat), fut.reshape(-1, T, feat), fut_in.reshape(-1, T, feat))
    return (enc.transpose(1, 0, 2, 3), fut.transpose(1, 0, 2, 3), fut_in.transpose(1, 0, 2, 3))


def _per_video_seconds(video):
    fps = cfg.fps
    db = np.stack((video["x"], video["y"], video["z"]), axis=-1)
    db = cut_head_or_tail_less_than_1sec(db, fps)
    return db.reshape(db.shape[0], db.shape[1] // fps, 3 * fps)


def _pad_others(oth, num_user):
    """Duplicate random others up to num_user-1 (np.random, as the reference) or truncate."""
    if oth.shape[0] < num_user - 1:
        n_real = oth.shape[0]
        extra = [oth]
        for _ in range(n_real, num_user - 1):
            extra.append(oth[np.random.randint(n_real)][np.newaxis])
        oth = np.concatenate(extra, axis=0)
    elif oth.shape[0] > num_user - 1:
        oth = oth[:num_user - 1]
    assert oth.shape[0] == num_user - 1
    return oth


def get_data(datadb, pick_user=False, num_user=48, verbose=False):
    """datadb: {video: {'x','y','z': (n_user, n_frame)}}.

    pick_user=False -> (enc, future, future_input), each (N, T, 90), users and videos pooled.
    pick_user=True  -> those three for the target user plus three (num_user-1, N, T, 90) arrays
    for the other users; every user of every video takes a turn as the target.
    """
    T = cfg.running_length
    tar = [[], [], []]
    oth = [[], [], []]
    for vid in datadb.keys():
        secs = _per_video_seconds(datadb[vid])
        if secs.shape[1] < 2 * T:
            if verbose:
                print("video %s only has %d seconds. skip..." % (vid, secs.shape[1]))
            continue
        if not pick_user:
            for dst, arr in zip(tar, reshape2second_stacks(secs, collapse_user=True)):
                dst.append(arr)
            continue
        for target in range(secs.shape[0]):
            others = _pad_others(np.delete(secs, target, axis=0), num_user)
            for dst, arr in zip(tar, reshape2second_stacks(secs[target][np.newaxis], collapse_user=True)):
                dst.append(arr)
            for dst, arr in zip(oth, reshape2second_stacks(others, collapse_user=False)):
                dst.append(arr)
    feat = 3 * cfg.fps
    cat = lambda lst, axis, empty: np.concatenate(lst, axis=axis) if lst else np.zeros(empty)
    out = [cat(a, 0, (0, T, feat)) for a in tar]
    if pick_user:
        out += [cat(a, 1, (num_user - 1, 0, T, feat)) for a in oth]
    if cfg.time_shift:
        out[0] = out[0][:, :-1]
        if pick_user:
            out[3] = out[3][:, :, :-1]
    return tuple(out)


def get_gt_target_xyz(y):
    """(N,T,90) interleaved xyz or (N,T,30,3) -> (N,T,6) = [mean xyz, population variance xyz]."""
    if y.shape[-1] == 3:
        assert y.ndim == 4
        comps = [y[:, :, :, a] for a in range(3)]
    else:
        assert y.ndim == 3 and y.shape[-1] % 3 == 0
        comps = [y[:, :, a::3] for a in range(3)]
    stats = [np.mean(c, axis=-1)[:, :, np.newaxis] for c in comps]
    stats += [np.var(c, axis=-1)[:, :, np.newaxis] for c in comps]
    return np.concatenate(stats, axis=-1)


def get_gt_target_xyz_oth(y):
    """(N,T,U,30,3) -> (N,T,U,6)."""
    assert y.ndim == 5 and y.shape[-1] == 3
    comps = [y[..., a] for a in range(3)]
    stats = [np.mean(c, axis=-1)[..., np.newaxis] for c in comps]
    stats += [np.var(c, axis=-1)[..., np.newaxis] for c in comps]
    return np.concatenate(stats, axis=-1)


def slice_layer(dimension, start, end):
    """Callable cropping `dimension` of an array to [start, end) (a Keras Lambda in the reference)."""
    def func(x):
        index = [slice(None)] * x.ndim
        index[dimension] = slice(start, end)
        return x[tuple(index)]
    return func


def reshape_others_data(video_db_oth):
    """(U-1, N, T, 90) -> (N, T, U-1, 30, 3), the layout of the model's `others` input."""
    a = video_db_oth.transpose((1, 2, 0, 3))
    return a.reshape(a.shape[0], a.shape[1], a.shape[2], cfg.fps, 3)


def get_shuffle_index(data_length, rng=None):
    rng = rng or np.random
    idx = np.arange(data_length)
    rng.shuffle(idx)
    return idx


def shuffle_data(index_shuf, arr):
    return np.asarray(arr)[np.asarray(index_shuf)]


def generator_train2(datadb, phase="train", num_user=34, video_keys=None):
    """Endless generator of ([enc, others, dec_in], target) minibatches, one (video, target user)
    at a time, batch = cfg.batch_size with a short last slice (reference generator_train2).

    enc (b,T,90) or (b,T,6) with cfg.input_mean_var; others (b,T,U-1,6) [or raw (b,T,U-1,90)
    without input_mean_var]; dec_in (b,1,*) = last encoder second; target (b,T,6) with
    cfg.predict_mean_var else raw (b,T,90); phase 'test' yields the raw future (b,T,1,30,3).
    """
    fps = cfg.fps
    T = cfg.running_length
    keys = list(datadb.keys()) if video_keys is None else list(video_keys)
    usable = [k for k in keys if _per_video_seconds(datadb[k]).shape[1] >= 2 * T]
    if not usable:
        raise ValueError("no video has at least %d seconds" % (2 * T))
    ii = 0
    while True:
        secs = _per_video_seconds(datadb[usable[ii % len(usable)]])
        for target in range(secs.shape[0]):
            others = _pad_others(np.delete(secs, target, axis=0), num_user)
            tar, tar_fut, _ = reshape2second_stacks(secs[target][np.newaxis], collapse_user=True)
            _, oth_fut, _ = reshape2second_stacks(others, collapse_user=False)
            tar4 = tar.reshape(tar.shape[0], T, fps, 3)
            fut4 = tar_fut.reshape(tar_fut.shape[0], T, fps, 3)
            oth5 = reshape_others_data(oth_fut)
            if cfg.input_mean_var:
                enc = get_gt_target_xyz(tar4)
                oth_in = get_gt_target_xyz_oth(oth5)
            else:
                enc = tar
                oth_in = oth5.transpose((0, 1, 2, 3, 4)).reshape(oth5.shape[0], T, oth5.shape[2], -1)
            dec_in = enc[:, -1, :][:, np.newaxis, :]
            if phase == "test":
                target_data = fut4[:, :, np.newaxis, :, :]
            elif cfg.predict_mean_var:
                target_data = get_gt_target_xyz(fut4)
            else:
                target_data = tar_fut
            bs = cfg.batch_size
            for lo in range(0, enc.shape[0], bs):
                yield [enc[lo:lo + bs], oth_in[lo:lo + bs], dec_in[lo:lo + bs]], target_data[lo:lo + bs]
        ii += 1


# ---------------------------------------------------------------------------------------------
# Pickled interchange with the reference's scripts (§8(f) rank 4): the dataset dictionaries they read
# and the prediction / ground-truth lists their plotting and evaluation scripts consume.
# ---------------------------------------------------------------------------------------------
def load_datadb(path):
    """`pickle.load(open(path,'rb'), encoding='latin1')` as given_others...py:338-340 / lstm_keras.py:84 do (the
    Shanghai / Tsinghua files were written by Python 2): dict[video] -> {'x','y','z': (n_user, n_frame)}; then
    clip_xyz, as every script does right after loading."""
    import pickle
    with open(path, "rb") as f:
        return clip_xyz(pickle.load(f, encoding="latin1"))


def save_decoded_sentences(decoded_sentence_list, gt_sentence_list, tag="", directory="."):
    """The two files every test loop of the reference ends with (FoV_seq2seq.py:193-194, given_others...py:699-700,
    lstm_keras.py:203-204): 'decoded_sentence<tag>.p' = list (one entry per batch) of predictions, and
    'gt_sentence_list<tag>.p' = list of the raw ground-truth futures, plain pickles of NumPy arrays."""
    import os
    import pickle
    paths = (os.path.join(directory, "decoded_sentence%s.p" % tag), os.path.join(directory, "gt_sentence_list%s.p" % tag))
    for p, obj in zip(paths, (decoded_sentence_list, gt_sentence_list)):
        with open(p, "wb") as f:
            pickle.dump([np.asarray(a) for a in obj], f)
    return paths
