"""Torch-tensor wrappers over the C ABI (include/fov360.h).

PyTorch is plumbing only: device memory, streams and (elsewhere) torch.distributed.  Every
function here hands raw device pointers and sizes to libfov360_hip.so; nothing is computed by
torch and there is no fallback path.
"""
import os

import torch

from . import _lib
from ._lib import (ACT_HARD_SIGMOID, ACT_SIGMOID, IMPL_AUTO, IMPL_CLUSTER, IMPL_GENERIC,  # noqa: F401
                   FovError, check)

_ACT = {"sigmoid": ACT_SIGMOID, "hard_sigmoid": ACT_HARD_SIGMOID, ACT_SIGMOID: ACT_SIGMOID,
        ACT_HARD_SIGMOID: ACT_HARD_SIGMOID}
_IMPL = {"auto": IMPL_AUTO, "generic": IMPL_GENERIC, "cluster": IMPL_CLUSTER, IMPL_AUTO: IMPL_AUTO,
         IMPL_GENERIC: IMPL_GENERIC, IMPL_CLUSTER: IMPL_CLUSTER}


def act_code(act):
    return _ACT[act]


def impl_code(impl):
    return _IMPL[impl]


def _dev(t, name):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous float32 tensor on the GPU" % name)
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current stream's handle.  torch.cuda.current_stream() builds a Stream object through several Python layers (8 us
    under a profiler, once per launch); the two C entry points underneath answer in well under a microsecond."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_ENV_KNOBS = ("FOV_FORCE_SAFE_EXCHANGE", "FOV_TWO_LAUNCHES", "FOV_DBG_RESIDENT_LIMIT", "FOV_NO_CELL_PATCH", "FOV_NO_CONV_PATCH", "FOV_NO_WIDE16", "FOV_BWD_STEPPED",
              "FOV_NO_WGRAD_FUSION", "FOV_NO_DX_FUSION", "FOV_BWD_GROUPS4", "FOV_GEMM_BF16_NOREMAP", "FOV_GEMM_BF16_SPLIT", "FOV_GEMM_BF16_SHALLOW", "FOV_NO_WGRAD_GROUP", "FOV_NO_WIDE16_TRIO", "FOV_DBG_TRACE", "FOV_GEMM_VARIANT",
              "FOV_GEMM_SPLIT", "FOV_NO_XCD_PAD", "FOV_XCD_PAD_MAX", "FOV_NO_BWD16_NARROW", "FOV_BWD16_GROUPS", "FOV_NO_STACK2", "FOV_PAIR", "FOV_NO_WGRAD_LINES")
_env_seen = None
# os.environ.get costs 0.7 us per key (encode, lookup, decode): 9 us for the knob list, paid at every workspace / scratch
# fetch - 0.15 ms of a 0.3 ms training step at batch 32.  The mapping underneath (bytes -> bytes on POSIX) answers the
# same question in 0.4 us; where it is absent the portable path runs.
_ENV_DATA = getattr(os.environ, "_data", None)
_ENV_KEYS_B = tuple(os.fsencode(k) for k in _ENV_KNOBS)
if not isinstance(_ENV_DATA, dict) or (_ENV_DATA and not isinstance(next(iter(_ENV_DATA)), bytes)):
    _ENV_DATA = None
_FORCE_SAFE_KEY_B = os.fsencode("FOV_FORCE_SAFE_EXCHANGE")


def _env_snapshot():
    if _ENV_DATA is not None:
        return tuple(map(_ENV_DATA.get, _ENV_KEYS_B))
    return tuple(os.environ.get(k) for k in _ENV_KNOBS)


def _sync_env():
    """The library caches its environment knobs (no getenv on a launch path): tell it when one of them changed."""
    global _env_seen
    cur = _env_snapshot()
    if cur != _env_seen:
        if _env_seen is not None or any(v is not None for v in cur):
            _lib.lib().fov_reload_env()
        _env_seen = cur


def _apply_force_safe(holder):
    """FOV_FORCE_SAFE_EXCHANGE=1 (read per call: the tests flip it between calls) -> the workspace's header word that keeps
    every exchanging kernel on the placement-independent granule exchange (fov_workspace_force_safe)."""
    want = (_ENV_DATA.get(_FORCE_SAFE_KEY_B) == b"1") if _ENV_DATA is not None else os.environ.get("FOV_FORCE_SAFE_EXCHANGE", "") == "1"
    _sync_env()
    if holder.buf is not None and getattr(holder, "_forced", (None, False)) != (holder.buf.data_ptr(), want):
        if want or getattr(holder, "_forced", (None, False))[1]:
            with torch.cuda.device(holder.buf.device):
                check(_lib.lib().fov_workspace_force_safe(holder.buf.data_ptr(), holder.buf.numel(), 1 if want else 0, _stream()))
        holder._forced = (holder.buf.data_ptr(), want)


def _new_workspace(nbytes, device):
    """A fresh workspace buffer, zero-filled ONCE through the C ABI (fov_workspace_init): the persistent kernels keep
    a header and monotone epoch tags in it across calls, no call clears anything (include/fov360.h, Conventions)."""
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    with torch.cuda.device(buf.device):
        check(_lib.lib().fov_workspace_init(buf.data_ptr(), buf.numel(), _stream()))
    return buf


class Workspace:
    """Caller-owned, stateful scratch of the persistent kernels (grown on demand, reused across calls, one stream at
    a time).  Its sticky timeout word makes a failed launch poison every later one until check() reports it."""

    def __init__(self, device=None):
        self.device = device
        self.buf = None

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = _new_workspace(nbytes, device)
        _apply_force_safe(self)
        return self.buf

    def check(self):
        """Synchronise and raise FovError(ERR_TIMEOUT) if a bounded in-kernel wait has given up since the last check
        (the failure is cleared by reporting it)."""
        if self.buf is not None:
            check(_lib.lib().fov_check_status(self.buf.data_ptr(), self.buf.numel(), _stream()))


    def exchange_mode(self):
        """1 = same-XCD fast exchange, 2 = placement-independent exchange (diagnostic)."""
        return _lib.lib().fov_exchange_mode(self.buf.data_ptr(), self.buf.numel(), _stream())


_default_ws = {}


def default_workspace(device):
    key = (device.type, device.index)
    if key not in _default_ws:
        _default_ws[key] = Workspace(device)
    return _default_ws[key]


def lstm_seq(x, K, R, b, h0=None, c0=None, act="sigmoid", impl="auto", return_sequences=True, workspace=None):
    """keras LSTM(return_sequences, return_state)(x, initial_state=[h0, c0]) -> (hs|None, hT, cT)."""
    x, K, R, b = _dev(x, "x"), _dev(K, "K"), _dev(R, "R"), _dev(b, "b")
    h0, c0 = _dev(h0, "h0"), _dev(c0, "c0")
    B, T, F = x.shape
    H = R.shape[0]
    assert K.shape == (F, 4 * H) and R.shape == (H, 4 * H) and b.shape == (4 * H,)
    hs = torch.empty((B, T, H), dtype=torch.float32, device=x.device) if return_sequences else None
    hT = torch.empty((B, H), dtype=torch.float32, device=x.device)
    cT = torch.empty((B, H), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    impl = impl_code(impl)
    ws = (workspace or default_workspace(x.device))
    buf = ws.get(L.fov_lstm_seq_workspace_bytes(B, T, F, H, impl), x.device)
    check(L.fov_lstm_seq_fwd(_ptr(x), _ptr(K), _ptr(R), _ptr(b), _ptr(h0), _ptr(c0), _ptr(hs), _ptr(hT), _ptr(cT),
                             B, T, F, H, act_code(act), impl, buf.data_ptr(), buf.numel(), _stream()))
    return hs, hT, cT


def dense(x, W, b, activation="tanh", out=None):
    x, W, b = _dev(x, "x"), _dev(W, "W"), _dev(b, "b")
    lead = x.shape[:-1]
    In, Out = W.shape
    x2 = x.reshape(-1, In)
    y = torch.empty((x2.shape[0], Out), dtype=torch.float32, device=x.device) if out is None else _dev(out, "out")
    assert y.numel() == x2.shape[0] * Out
    check(_lib.lib().fov_dense_fwd(_ptr(x2), _ptr(W), _ptr(b), _ptr(y), x2.shape[0], In, Out,
                                   1 if activation == "tanh" else 0, _stream()))
    return y.reshape(*lead, Out)


_W_ORDER = ("enc_K", "enc_R", "enc_b", "dec_K", "dec_R", "dec_b", "dense_W", "dense_b")


def seq2seq_decode(enc_in, dec_in0, w, T_out, act="sigmoid", impl="auto", workspace=None, out=None, hT=None, cT=None):
    """Fused encoder + autoregressive decoder (FoV_seq2seq.py:154-178, batched) -> (B,T_out,F_dec); hT / cT (B,H): the decoder's
    final state, if wanted."""
    enc_in, dec_in0 = _dev(enc_in, "enc_in"), _dev(dec_in0, "dec_in0")
    ws_t = [_dev(w[k], k) for k in _W_ORDER]
    B, T_in, F_enc = enc_in.shape
    H = ws_t[1].shape[0]
    F_dec = ws_t[6].shape[1]
    assert dec_in0.shape == (B, 1, F_dec)
    if out is None:
        out = torch.empty((B, T_out, F_dec), dtype=torch.float32, device=enc_in.device)
    L = _lib.lib()
    impl = impl_code(impl)
    ws = (workspace or default_workspace(enc_in.device))
    buf = ws.get(L.fov_seq2seq_decode_workspace_bytes(B, T_in, T_out, F_enc, F_dec, H, impl), enc_in.device)
    check(L.fov_seq2seq_decode_fwd(_ptr(enc_in), _ptr(dec_in0), *[_ptr(t) for t in ws_t], _ptr(out),
                                   None if hT is None else _ptr(_dev(hT, "hT")), None if cT is None else _ptr(_dev(cT, "cT")),
                                   B, T_in, T_out, F_enc, F_dec, H, act_code(act), impl,
                                   buf.data_ptr(), buf.numel(), _stream()))
    return out


def seq2seq_decoder(dec_in0, h0, c0, w, T_out, act="sigmoid", impl="auto", workspace=None, out=None):
    """The decoder half alone from a given state (decoder_model.predict fed its own output, FoV_seq2seq.py:156-178)
    -> (B,T_out,F_dec)."""
    dec_in0, h0, c0 = _dev(dec_in0, "dec_in0"), _dev(h0, "h0"), _dev(c0, "c0")
    ws_t = [_dev(w[k], k) for k in _W_ORDER[3:]]
    B, H = h0.shape
    F_dec = ws_t[3].shape[1]
    assert dec_in0.shape == (B, 1, F_dec)
    if out is None:
        out = torch.empty((B, T_out, F_dec), dtype=torch.float32, device=h0.device)
    L = _lib.lib()
    impl = impl_code(impl)
    ws = (workspace or default_workspace(h0.device))
    buf = ws.get(L.fov_seq2seq_decode_workspace_bytes(B, 0, T_out, 1, F_dec, H, impl), h0.device)
    check(L.fov_seq2seq_decoder_fwd(_ptr(dec_in0), _ptr(h0), _ptr(c0), *[_ptr(t) for t in ws_t], _ptr(out), None, None,
                                    B, T_out, F_dec, H, act_code(act), impl, buf.data_ptr(), buf.numel(), _stream()))
    return out


def seq2seq_teacher_forced(enc_in, dec_in, w, act="sigmoid", impl="auto", workspace=None):
    """Training-graph forward (FoV_seq2seq.py:82-101) -> (B,T_out,F_dec)."""
    enc_in, dec_in = _dev(enc_in, "enc_in"), _dev(dec_in, "dec_in")
    ws_t = [_dev(w[k], k) for k in _W_ORDER]
    B, T_in, F_enc = enc_in.shape
    T_out, F_dec = dec_in.shape[1], dec_in.shape[2]
    H = ws_t[1].shape[0]
    out = torch.empty((B, T_out, F_dec), dtype=torch.float32, device=enc_in.device)
    L = _lib.lib()
    impl = impl_code(impl)
    ws = (workspace or default_workspace(enc_in.device))
    buf = ws.get(L.fov_seq2seq_tf_workspace_bytes(B, T_in, T_out, F_enc, F_dec, H, impl), enc_in.device)
    check(L.fov_seq2seq_tf_fwd(_ptr(enc_in), _ptr(dec_in), *[_ptr(t) for t in ws_t], _ptr(out),
                               B, T_in, T_out, F_enc, F_dec, H, act_code(act), impl,
                               buf.data_ptr(), buf.numel(), _stream()))
    return out


def meanvar_xyz(y, fps=30):
    """(..., 3*fps) or (..., fps, 3) device tensor -> (..., 6) = [mean xyz, population var xyz]."""
    y = _dev(y, "y")
    if y.shape[-1] == 3 and y.shape[-2] == fps:
        lead = y.shape[:-2]
    else:
        assert y.shape[-1] == 3 * fps
        lead = y.shape[:-1]
    rows = 1
    for d in lead:
        rows *= d
    out = torch.empty((*lead, 6), dtype=torch.float32, device=y.device)
    check(_lib.lib().fov_meanvar_xyz(_ptr(y), _ptr(out), rows, fps, _stream()))
    return out


# ---------------------------------------------------------------------------------------------
# training side (a6): forward with reserve, BPTT, Dense backward, MSE gradient, optimizers
# ---------------------------------------------------------------------------------------------
def bf16_layer_supported(F, H):
    return H == 256 and 1 <= F <= 256


def lstm_seq_bf16(x, K, R, b, h0=None, c0=None, act="sigmoid", workspace=None, out=None, reserve=True):
    """LSTM layer with bf16 matrix-core operands (configs[4]; fp32 tensors in and out, H = 256, F <= 256).
    `out` may carry preallocated (hs, hT, cT, reserve) tensors (None entries are not written).
    -> (hs, hT, cT, reserve)."""
    x, K, R, b = _dev(x, "x"), _dev(K, "K"), _dev(R, "R"), _dev(b, "b")
    h0, c0 = _dev(h0, "h0"), _dev(c0, "c0")
    B, T, F = x.shape
    H = R.shape[0]
    if out is None:
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
        out = (e(B, T, H), e(B, H), e(B, H), e(B, T, 5, H) if reserve else None)
    hs, hT, cT, res = out
    L = _lib.lib()
    ws = (workspace or default_workspace(x.device))
    buf = ws.get(L.fov_lstm_seq_workspace_bytes(B, T, 128, 256, IMPL_CLUSTER), x.device)
    check(L.fov_lstm_seq_fwd_bf16(_ptr(x), _ptr(K), _ptr(R), _ptr(b), _ptr(h0), _ptr(c0), _ptr(hs), _ptr(hT), _ptr(cT),
                                  _ptr(res), B, T, F, H, act_code(act), buf.data_ptr(), buf.numel(), _stream()))
    return hs, hT, cT, res


def lstm_stack2_bf16_supported(B, T, F, H):
    return os.environ.get("FOV_NO_STACK2", "") != "1" and bool(_lib.lib().fov_lstm_stack2_supported_bf16(B, T, F, H))


def lstm_stack2_bf16(x, layer1, layer2, act="sigmoid", workspace=None, out1=None, out2=None, reserve=True):
    """Two stacked bf16 LSTM layers (zero initial states) as ONE wavefront launch: layer = (K, R, b); out1 / out2 may carry
    preallocated (hs, hT, cT, reserve) of the layer (None entries are not written).  -> (out1, out2)."""
    x = _dev(x, "x")
    K1, R1, b1 = (_dev(t, "layer1") for t in layer1)
    K2, R2, b2 = (_dev(t, "layer2") for t in layer2)
    B, T, F = x.shape
    H = R1.shape[0]
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
    if out1 is None:
        out1 = (e(B, T, H), e(B, H), e(B, H), e(B, T, 5, H) if reserve else None)
    if out2 is None:
        out2 = (e(B, T, H), e(B, H), e(B, H), e(B, T, 5, H) if reserve else None)
    L = _lib.lib()
    ws = (workspace or default_workspace(x.device))
    buf = ws.get(L.fov_lstm_seq_workspace_bytes(B, T, 128, 256, IMPL_CLUSTER), x.device)
    check(L.fov_lstm_stack2_fwd_bf16(_ptr(x), _ptr(K1), _ptr(R1), _ptr(b1), _ptr(K2), _ptr(R2), _ptr(b2),
                                     *[_ptr(t) for t in out1], *[_ptr(t) for t in out2], B, T, F, H, act_code(act),
                                     buf.data_ptr(), buf.numel(), _stream()))
    return out1, out2


def lstm_stack2_supported(B, T, F, H):
    return os.environ.get("FOV_NO_STACK2", "") != "1" and bool(_lib.lib().fov_lstm_stack2_supported(B, T, F, H))


def lstm_stack2(x, layer1, layer2, state1=None, state2=None, act="sigmoid", workspace=None, reserve=False, final1=None, final2=None):
    """Two stacked fp32 LSTM layers (F <= 96 -> 512 -> 512) as ONE launch, layer 2 a few steps behind layer 1 on other CUs
    (fov_lstm_stack2_fwd).  layer = (K, R, b); state = (h0, c0) or None; final = (hT, cT) tensors the layer's final state is
    written into (e.g. views of a carried-state buffer) or None.  -> ((hs, hT, cT, reserve), (hs, hT, cT, reserve))."""
    x = _dev(x, "x")
    K1, R1, b1 = (_dev(t, "layer1") for t in layer1)
    K2, R2, b2 = (_dev(t, "layer2") for t in layer2)
    h01, c01 = (None, None) if state1 is None else (_dev(state1[0], "h0"), _dev(state1[1], "c0"))
    h02, c02 = (None, None) if state2 is None else (_dev(state2[0], "h0"), _dev(state2[1], "c0"))
    B, T, F = x.shape
    H = R1.shape[0]
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
    f1 = (e(B, H), e(B, H)) if final1 is None else (_dev(final1[0], "hT"), _dev(final1[1], "cT"))
    f2 = (e(B, H), e(B, H)) if final2 is None else (_dev(final2[0], "hT"), _dev(final2[1], "cT"))
    out1 = (e(B, T, H), f1[0], f1[1], e(B, T, 5, H) if reserve else None)
    out2 = (e(B, T, H), f2[0], f2[1], e(B, T, 5, H) if reserve else None)
    L = _lib.lib()
    ws = (workspace or default_workspace(x.device))
    buf = ws.get(L.fov_lstm_seq_workspace_bytes(B, T, F, H, IMPL_AUTO), x.device)
    check(L.fov_lstm_stack2_fwd(_ptr(x), _ptr(K1), _ptr(R1), _ptr(b1), _ptr(h01), _ptr(c01), _ptr(K2), _ptr(R2), _ptr(b2),
                                _ptr(h02), _ptr(c02), *[_ptr(t) for t in out1], *[_ptr(t) for t in out2], B, T, F, H,
                                act_code(act), buf.data_ptr(), buf.numel(), _stream()))
    return out1, out2


def lstm_seq_train(x, K, R, b, h0=None, c0=None, act="sigmoid", impl="auto", workspace=None, out=None, dtype="f32"):
    """Forward that also returns the reserve (B,T,5,H).  `out` may carry preallocated
    (hs, hT, cT, reserve) tensors.  -> (hs, hT, cT, reserve).  dtype 'bf16': bf16 matrix-core operands."""
    if dtype == "bf16":
        return lstm_seq_bf16(x, K, R, b, h0, c0, act=act, workspace=workspace, out=out)
    x, K, R, b = _dev(x, "x"), _dev(K, "K"), _dev(R, "R"), _dev(b, "b")
    h0, c0 = _dev(h0, "h0"), _dev(c0, "c0")
    B, T, F = x.shape
    H = R.shape[0]
    if out is None:
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
        out = (e(B, T, H), e(B, H), e(B, H), e(B, T, 5, H))
    hs, hT, cT, res = out      # hT / cT may be None (not written)
    L = _lib.lib()
    impl = impl_code(impl)
    ws = (workspace or default_workspace(x.device))
    buf = ws.get(L.fov_lstm_seq_workspace_bytes(B, T, F, H, impl), x.device)
    check(L.fov_lstm_seq_fwd_train(_ptr(x), _ptr(K), _ptr(R), _ptr(b), _ptr(h0), _ptr(c0), _ptr(hs), _ptr(hT),
                                   _ptr(cT), _ptr(res), B, T, F, H, act_code(act), impl, buf.data_ptr(),
                                   buf.numel(), _stream()))
    return hs, hT, cT, res


class Scratch:
    """Second caller-owned buffer for the training kernels (split-K partials, dh/dc carries)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = _new_workspace(nbytes, device)
        _apply_force_safe(self)
        return self.buf

    def check(self):
        """Raise FovError(ERR_TIMEOUT) if the persistent BPTT kernel gave up a bounded wait."""
        if self.buf is not None:
            check(_lib.lib().fov_check_status(self.buf.data_ptr(), self.buf.numel(), _stream()))


_default_scratch = Scratch()
# fov_lstm_seq_bwd keeps the persistent kernel's header and granule area at the head of its workspace (stateful, zero-filled
# once): its default buffer is never shared with the calls that use theirs for split partials
_default_bwd_scratch = Scratch()


def lstm_seq_bwd(x, K, R, hs, reserve, h0=None, c0=None, dhs=None, dhT=None, dcT=None, dK=None, dR=None, db=None,
                 need_dx=False, need_state_grads=False, act="sigmoid", accumulate=False, dz=None, scratch=None,
                 need_weight_grads=True, dtype="f32"):
    """BPTT of one layer -> dict(dz, dx, dK, dR, db, dh0, dc0).  need_weight_grads=False computes the data path
    only (dz, dx, state gradients): a caller that walks a decoder step by step stacks dz and forms the
    weight gradients once, over all steps."""
    x, K, R, hs, reserve = _dev(x, "x"), _dev(K, "K"), _dev(R, "R"), _dev(hs, "hs"), _dev(reserve, "reserve")
    B, T, F = x.shape
    H = R.shape[0]
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
    dz = e(B, T, 4 * H) if dz is None else dz
    if need_weight_grads:
        dK = e(F, 4 * H) if dK is None else dK
        dR = e(H, 4 * H) if dR is None else dR
        db = e(4 * H) if db is None else db
    else:
        dK = dR = db = None
    dx = e(B, T, F) if need_dx else None
    dh0 = e(B, H) if need_state_grads else None
    dc0 = e(B, H) if need_state_grads else None
    L = _lib.lib()
    buf = (scratch or _default_bwd_scratch).get(L.fov_lstm_seq_bwd_workspace_bytes(B, T, F, H), x.device)
    bwd = L.fov_lstm_seq_bwd_bf16 if dtype == "bf16" else L.fov_lstm_seq_bwd
    check(bwd(_ptr(x), _ptr(K), _ptr(R), _ptr(_dev(h0, "h0")), _ptr(_dev(c0, "c0")), _ptr(hs),
              _ptr(reserve), _ptr(_dev(dhs, "dhs")), _ptr(_dev(dhT, "dhT")), _ptr(_dev(dcT, "dcT")),
              _ptr(dz), _ptr(dx), _ptr(dK), _ptr(dR), _ptr(db), _ptr(dh0), _ptr(dc0),
              B, T, F, H, act_code(act), 1 if accumulate else 0, buf.data_ptr(), buf.numel(), _stream()))
    return {"dz": dz, "dx": dx, "dK": dK, "dR": dR, "db": db, "dh0": dh0, "dc0": dc0}


def lstm_stack2_bwd_supported(B, T, F, H):
    return bool(_lib.lib().fov_lstm_stack2_bwd_supported(B, T, F, H))


def lstm_stack2_bwd(x, layer1, layer2, tape1, tape2, dhs2=None, dhT2=None, dcT2=None, dhT1=None, dcT1=None, grads1=None, grads2=None,
                    need_state_grads=False, act="sigmoid", accumulate=False, scratch=None):
    """BPTT of two stacked width-512 layers in one launch (fov_lstm_stack2_bwd).  layer1 = (K1, R1), layer2 = (K2, R2);
    tape = (hs, reserve, h0, c0) of the training forward; grads = (dK, dR, db) tensors or None (data path only).
    -> dict(dz1, dz2, dh0_1, dc0_1, dh0_2, dc0_2)."""
    x = _dev(x, "x")
    B, T, F = x.shape
    R1, K2, R2 = _dev(layer1[1], "R1"), _dev(layer2[0], "K2"), _dev(layer2[1], "R2")
    H = R1.shape[0]
    hs1, res1, h01, c01 = tape1
    hs2, res2, h02, c02 = tape2
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
    dz1, dz2 = e(B, T, 4 * H), e(B, T, 4 * H)
    st = [e(B, H) if need_state_grads else None for _ in range(4)]
    g1 = grads1 if grads1 is not None else (None, None, None)
    g2 = grads2 if grads2 is not None else (None, None, None)
    L = _lib.lib()
    buf = (scratch or _default_bwd_scratch).get(L.fov_lstm_stack2_bwd_workspace_bytes(B, T, F, H), x.device)
    check(L.fov_lstm_stack2_bwd(_ptr(x), _ptr(R1), _ptr(K2), _ptr(R2), _ptr(_dev(h01, "h0_1")), _ptr(_dev(c01, "c0_1")),
                                _ptr(_dev(h02, "h0_2")), _ptr(_dev(c02, "c0_2")), _ptr(_dev(hs1, "hs1")), _ptr(_dev(res1, "reserve1")),
                                _ptr(_dev(hs2, "hs2")), _ptr(_dev(res2, "reserve2")), _ptr(_dev(dhs2, "dhs2")), _ptr(_dev(dhT2, "dhT2")),
                                _ptr(_dev(dcT2, "dcT2")), _ptr(_dev(dhT1, "dhT1")), _ptr(_dev(dcT1, "dcT1")), _ptr(dz1), _ptr(dz2),
                                _ptr(g1[0]), _ptr(g1[1]), _ptr(g1[2]), _ptr(g2[0]), _ptr(g2[1]), _ptr(g2[2]), _ptr(st[0]), _ptr(st[1]),
                                _ptr(st[2]), _ptr(st[3]), B, T, F, H, act_code(act), 1 if accumulate else 0, buf.data_ptr(), buf.numel(),
                                _stream()))
    return {"dz1": dz1, "dz2": dz2, "dh0_1": st[0], "dc0_1": st[1], "dh0_2": st[2], "dc0_2": st[3]}


def lstm_seq_wgrad(x, hs, dz, dK=None, dR=None, db=None, h0=None, accumulate=False, scratch=None, dtype="f32"):
    """The weight-gradient half of lstm_seq_bwd from the dz tape a `need_weight_grads=False` call left: dK = x^T dz,
    dR = h_{t-1}^T dz, db = colsum(dz), bit-identical to the single call.  `scratch` must not be the workspace of a BPTT kernel
    that may run concurrently on another stream."""
    x, hs, dz = _dev(x, "x"), _dev(hs, "hs"), _dev(dz, "dz")
    B, T, F = x.shape
    H = hs.shape[-1]
    assert hs.shape == (B, T, H) and dz.shape == (B, T, 4 * H)
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(L.fov_lstm_seq_bwd_workspace_bytes(B, T, F, H), x.device)
    check(L.fov_lstm_seq_wgrad(_ptr(x), _ptr(hs), _ptr(_dev(h0, "h0")), _ptr(dz), _ptr(dK), _ptr(dR), _ptr(db), B, T, F, H,
                               1 if accumulate else 0, 1 if dtype == "bf16" else 0, buf.data_ptr(), buf.numel(), _stream()))
    return {"dK": dK, "dR": dR, "db": db}


def lstm_seq_wgrad_pair_one_launch(B, T1, T2, H):
    return bool(_lib.lib().fov_lstm_seq_wgrad_pair_one_launch(B, T1, T2, H))


def lstm_seq_wgrad_pair(layer1, layer2, accumulate=False, scratch=None):
    """All weight gradients of an encoder / decoder pair from the dz tapes of two `need_weight_grads=False` BPTT calls
    (fov_lstm_seq_wgrad_pair).  layer = (x (B,T,F), hs (B,T,H), h0 or None, dz (B,T,4H), dK, dR, db)."""
    a = [_dev(layer1[0], "x1"), _dev(layer1[1], "hs1"), _dev(layer1[2], "h0_1"), _dev(layer1[3], "dz1")]
    b = [_dev(layer2[0], "x2"), _dev(layer2[1], "hs2"), _dev(layer2[2], "h0_2"), _dev(layer2[3], "dz2")]
    B, T1, F1 = a[0].shape
    B2, T2, F2 = b[0].shape
    H = a[1].shape[-1]
    assert B == B2 and a[1].shape == (B, T1, H) and b[1].shape == (B, T2, H) and a[3].shape == (B, T1, 4 * H) and b[3].shape == (B, T2, 4 * H)
    L = _lib.lib()
    need = max(L.fov_lstm_seq_bwd_workspace_bytes(B, T1, F1, H), L.fov_lstm_seq_bwd_workspace_bytes(B, T2, F2, H))
    buf = (scratch or _default_scratch).get(need, a[0].device)
    check(L.fov_lstm_seq_wgrad_pair(_ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(layer1[4]), _ptr(layer1[5]), _ptr(layer1[6]), T1, F1,
                                    _ptr(b[0]), _ptr(b[1]), _ptr(b[2]), _ptr(b[3]), _ptr(layer2[4]), _ptr(layer2[5]), _ptr(layer2[6]), T2, F2,
                                    B, H, 1 if accumulate else 0, buf.data_ptr(), buf.numel(), _stream()))


_side_streams = {}


def side_stream(device, priority=1, owner=None):
    """THE HIP stream of the given priority (> 0 low, 0 normal, < 0 high) of `device`, created on that device (fov_stream_create
    makes it on the current HIP device) on first use and wrapped as a torch stream object.  One stream per (device, priority)
    for the life of the process: every trainer that wants low-priority side work shares it - nothing leaks per trainer, and
    nothing is ever destroyed (torch's caching allocator keeps blocks and events bound to a stream it has seen; destroying the
    handle under it - tried in round 5 from a finalizer - crashed the interpreter at exit)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(priority))
    st = _side_streams.get(key)
    if st is None:
        h = _ct.c_void_p()
        with torch.cuda.device(key[0]):
            check(_lib.lib().fov_stream_create(int(priority), _ct.byref(h)))
            st = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", key[0]))
        _side_streams[key] = st
    return st


def act_bwd(dy, y, base=None, activation="tanh", out=None):
    """out = base + dy * act'(y)  (tanh: 1 - y^2; relu: [y > 0]; linear: 1)."""
    dy, y = _dev(dy, "dy"), _dev(y, "y")
    assert dy.shape == y.shape
    out = torch.empty_like(y) if out is None else out
    check(_lib.lib().fov_act_bwd(_ptr(dy), _ptr(y), _ptr(_dev(base, "base")), _ptr(out), y.numel(),
                                 _ACT_CODES.get(activation, 0), _stream()))
    return out


_ACT_CODES = {None: 0, "linear": 0, "tanh": 1, "relu": 2, "exp": 3}


def act_fwd(x, activation, out=None):
    """y = act(x) elementwise: 'tanh' | 'relu' | 'exp' | None (in place when out is x)."""
    x = _dev(x, "x")
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().fov_act_fwd(_ptr(x), _ptr(out), x.numel(), _ACT_CODES[activation], _stream()))
    return out


def tf_head_supported(B, H, M, O):
    return bool(_lib.lib().fov_tf_head_supported(B, H, M, O))


def tf_head_fwd(h, w):
    """lstm.py:321-337, both heads in one launch: w has mu_W1, mu_b1, mu_W2, mu_b2, var_W1, ... -> (a1, mu, a3, var)."""
    h = _dev(h, "h")
    B, H = h.shape
    M, O = w["mu_W2"].shape
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=h.device)
    a1, mu, a3, var = e(B, M), e(B, O), e(B, M), e(B, O)
    names = ("mu_W1", "mu_b1", "mu_W2", "mu_b2", "var_W1", "var_b1", "var_W2", "var_b2")
    check(_lib.lib().fov_tf_head_fwd(_ptr(h), *[_ptr(_dev(w[k], k)) for k in names], _ptr(a1), _ptr(mu), _ptr(a3), _ptr(var),
                                     B, H, M, O, _stream()))
    return a1, mu, a3, var


def tf_head_bwd(h, w, head, dmu, dvar, g, accumulate=False):
    """Backward of tf_head_fwd in one launch: head = (a1, mu, a3, var); the eight gradients go to g[name] (added when
    accumulate); returns dh (B,H)."""
    h = _dev(h, "h")
    B, H = h.shape
    M, O = w["mu_W2"].shape
    a1, mu, a3, var = head
    dh = torch.empty((B, H), dtype=torch.float32, device=h.device)
    gn = ("mu_W1", "mu_b1", "mu_W2", "mu_b2", "var_W1", "var_b1", "var_W2", "var_b2")
    check(_lib.lib().fov_tf_head_bwd(_ptr(h), _ptr(w["mu_W1"]), _ptr(w["mu_W2"]), _ptr(w["var_W1"]), _ptr(w["var_W2"]), _ptr(a1),
                                     _ptr(mu), _ptr(a3), _ptr(var), _ptr(_dev(dmu, "dmu")), _ptr(_dev(dvar, "dvar")),
                                     *[_ptr(_dev(g[k], k)) for k in gn], _ptr(dh), B, H, M, O, 1 if accumulate else 0, _stream()))
    return dh


import ctypes as _ct

_PTR4 = _ct.c_void_p * 4
_INT4 = _ct.c_int * 4
_INT5 = _ct.c_int * 5


def _mlp_tables(layers, x_width):
    """layers: [(W (Din,Dout), b (Dout), activation)] -> (dims, codes, W pointers, b pointers) as ctypes arrays."""
    L = len(layers)
    assert 1 <= L <= 4
    dims = [int(x_width)] + [int(W.shape[1]) for W, _, _ in layers]
    for l, (W, b, _) in enumerate(layers):
        _dev(W, "W%d" % l); _dev(b, "b%d" % l)
        assert W.shape[0] == dims[l] and b.shape == (dims[l + 1],)
    pad = [None] * (4 - L)
    return (_INT5(*(dims + [0] * (4 - L))), _INT4(*([_ACT_CODES[a] for _, _, a in layers] + [0] * (4 - L))),
            _PTR4(*([W.data_ptr() for W, _, _ in layers] + pad)), _PTR4(*([b.data_ptr() for _, b, _ in layers] + pad)))


def mlp_head_supported(B, dims):
    L = len(dims) - 1
    return 1 <= L <= 4 and bool(_lib.lib().fov_mlp_head_supported(B, L, _INT5(*(list(dims) + [0] * (4 - L)))))


def mlp_head_fwd(x, layers, masks=None, n_mix=0):
    """Fused chain of up to four Dense layers on a few rows (mlp_head.hip): the heads of mycode/lstm.py:147-174 and
    :377-400.  layers [(W, b, activation)]; masks [per layer (B,Dout) or None]; n_mix > 0: the last layer gets the
    mixture split [softmax n | 3n | exp 3n | tanh 3n] instead of its activation.  -> [output of every layer]."""
    x = _dev(x, "x")
    B, D0 = x.shape
    dims, codes, Wp, bp = _mlp_tables(layers, D0)
    L = len(layers)
    acts = [torch.empty((B, W.shape[1]), dtype=torch.float32, device=x.device) for W, _, _ in layers]
    mp = None
    if masks is not None and any(m is not None for m in masks):
        for l, m in enumerate(masks):
            assert m is None or _dev(m, "mask").shape == acts[l].shape
        mp = _PTR4(*([_ptr(m) for m in masks] + [None] * (4 - len(masks))))
    check(_lib.lib().fov_mlp_head_fwd(_ptr(x), Wp, bp, mp, _PTR4(*([a.data_ptr() for a in acts] + [None] * (4 - L))), dims, codes, L,
                                      1 if n_mix else 0, int(n_mix), B, _stream()))
    return acts


def mlp_head_bwd(x, layers, acts, dlast, gW, gb, masks=None, need_dx=True, accumulate=False, scratch=None):
    """Backward of mlp_head_fwd: dlast = d loss / d pre-activation of the last layer; gW / gb lists receive (accumulate: are
    added) the gradients; returns dx (B,D0) or None."""
    x, dlast = _dev(x, "x"), _dev(dlast, "dlast")
    B, D0 = x.shape
    dims, codes, Wp, _ = _mlp_tables(layers, D0)
    L = len(layers)
    assert dlast.shape == (B, layers[-1][0].shape[1])
    for l in range(L):
        assert _dev(gW[l], "gW").shape == layers[l][0].shape and _dev(gb[l], "gb").shape == layers[l][1].shape
    dx = torch.empty((B, D0), dtype=torch.float32, device=x.device) if need_dx else None
    mp = None
    if masks is not None and any(m is not None for m in masks):
        mp = _PTR4(*([_ptr(m) for m in masks] + [None] * (4 - len(masks))))
    lib = _lib.lib()
    buf = (scratch or _default_scratch).get(lib.fov_mlp_head_bwd_workspace_bytes(B, L, dims), x.device)
    pad = [None] * (4 - L)
    check(lib.fov_mlp_head_bwd(_ptr(x), Wp, mp, _PTR4(*([_ptr(_dev(a, "act")) for a in acts] + pad)), _ptr(dlast),
                               _PTR4(*([g.data_ptr() for g in gW] + pad)), _PTR4(*([g.data_ptr() for g in gb] + pad)), _ptr(dx), dims,
                               codes, L, B, 1 if accumulate else 0, buf.data_ptr(), buf.numel(), _stream()))
    return dx


_gmm_loss_ws = {}


def gmm3d_loss_grad(params, y, n_pts, scale, weight_by_pi=False, scratch=None):
    """costfunc.mixture_3d_gaussian_loss (cost.py:486-549) + gradient at the head's pre-activations.  params (B,10n) from
    mlp_head_fwd(n_mix=n); y (B, ...) whose rows START with the n_pts scored frames (x,y,z interleaved): (B,T_y,3*fps)
    under cfg.process_in_seconds (second 0), (B,T,3) per frame.  -> (loss (1,), dpre (B,10n))."""
    params, y = _dev(params, "params"), _dev(y, "y")
    B = params.shape[0]
    n = params.shape[1] // 10
    assert params.shape[1] == 10 * n and y.shape[0] == B
    ldy = y.numel() // max(B, 1)
    loss = (torch.empty if B > 0 else torch.zeros)(1, dtype=torch.float32, device=y.device)
    dpre = torch.empty_like(params)
    # stateful workspace (ticket word, zero between calls): its own zero-filled buffer, never the shared scratch other calls scribble on
    key = (y.device.index, _stream())
    buf = _gmm_loss_ws.get(key)
    if buf is None or buf.numel() < 256 + 4 * B:
        buf = _gmm_loss_ws[key] = torch.zeros(max(4096, 256 + 4 * B), dtype=torch.uint8, device=y.device)
    check(_lib.lib().fov_gmm3d_loss_grad(_ptr(params), _ptr(y), ldy, _ptr(loss), _ptr(dpre), B, n, int(n_pts), float(scale),
                                         1 if weight_by_pi else 0, buf.data_ptr(), buf.numel(), _stream()))
    return loss, dpre


def gmm3d_sample(params, u, z, out=None):
    """One draw per frame from the 3-D mixture (utility.sample_mixture_3D's documented intent, utility.py:178-208): u (B,P)
    uniform picks the component by inverse CDF over pi, z (B,P,3) normal -> mu_m + chol(Sigma'_m) z.  out: (B,3P) rows,
    possibly a strided slot of a window; -> out."""
    params, u, z = _dev(params, "params"), _dev(u, "u"), _dev(z, "z")
    B, P = u.shape
    n = params.shape[1] // 10
    assert params.shape == (B, 10 * n) and z.shape == (B, P, 3)
    if out is None:
        out = torch.empty((B, 3 * P), dtype=torch.float32, device=u.device)
    ptr, ld = _row_view(out, 3 * P)
    check(_lib.lib().fov_gmm3d_sample(_ptr(params), _ptr(u), _ptr(z), ptr, ld, B, n, P, _stream()))
    return out


def gauss_nll_grad(mu, var, y, fps, scale, scratch=None):
    """Gaussian NLL of cost.py:190-229 -> (loss (1,), dmu (B,3), dvar (B,3)).  y (B,T_y,3*fps)."""
    mu, var, y = _dev(mu, "mu"), _dev(var, "var"), _dev(y, "y")
    B, T_y = y.shape[0], y.shape[1]
    assert mu.shape == (B, 3) and var.shape == (B, 3) and y.shape[2] == 3 * fps
    # the entry point overwrites the loss (an empty batch returns before it: zero then)
    loss = (torch.empty if B > 0 else torch.zeros)(1, dtype=torch.float32, device=y.device)
    dmu, dvar = torch.empty_like(mu), torch.empty_like(var)
    buf = (scratch or _default_scratch).get(4 * (B + 64), y.device)
    check(_lib.lib().fov_gauss_nll_grad(_ptr(mu), _ptr(var), _ptr(y), _ptr(loss), _ptr(dmu), _ptr(dvar), B, T_y, fps,
                                        float(scale), buf.data_ptr(), buf.numel(), _stream()))
    return loss, dmu, dvar


def categorical_crossentropy_grad(p, target, scratch=None, dp=None):
    """Keras categorical_crossentropy (TF backend, probabilities) over the last axis -> (dp like p, loss (1,))."""
    p, target = _dev(p, "p"), _dev(target, "target")
    assert p.shape == target.shape
    C = p.shape[-1]
    n_pix = p.numel() // C
    dp = torch.empty_like(p) if dp is None else dp
    loss = torch.zeros(1, dtype=torch.float32, device=p.device)
    buf = (scratch or _default_scratch).get(4 * ((n_pix + 255) // 256 + 64), p.device)
    check(_lib.lib().fov_categorical_crossentropy_grad(_ptr(p), _ptr(target), _ptr(dp), _ptr(loss), n_pix, C, buf.data_ptr(),
                                                       buf.numel(), _stream()))
    return dp, loss


def xyz_sum1_grad(p, dp, scratch=None):
    """cfg.add_xyz_sum1 term of costfunc._mse (cost.py:20-29) on p (..., C >= 3): adds its gradient into dp, returns the
    term as a (1,) tensor."""
    p, dp = _dev(p, "p"), _dev(dp, "dp")
    assert p.shape == dp.shape
    C = p.shape[-1]
    n_pix = p.numel() // C
    reg = torch.zeros(1, dtype=torch.float32, device=p.device)
    buf = (scratch or _default_scratch).get(4 * ((n_pix + 255) // 256 + 64), p.device)
    check(_lib.lib().fov_xyz_sum1_grad(_ptr(p), _ptr(dp), _ptr(reg), n_pix, C, buf.data_ptr(), buf.numel(), _stream()))
    return reg


def _row_view(x, width):
    """(B, width) view whose rows may be a slot of a wider window: -> (data_ptr, row stride in floats)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == width and x.stride(1) == 1
    return x.data_ptr(), (x.stride(0) if x.shape[0] > 1 else max(x.stride(0), width))


def sample_refeed(mu, var, noise, out=None, std="sqrt", planar=False):
    """One sampled second x = mu + sd(var) * noise (lstm.py:460-468 / lstm_keras.py:139-149).  mu, var (B,3); noise
    (B,3*fps) standard normal in the layout of x; `out` (B,3*fps) may be a row-strided slot of the input window."""
    mu, var, noise = _dev(mu, "mu"), _dev(var, "var"), _dev(noise, "noise")
    B, n = noise.shape
    out = torch.empty((B, n), dtype=torch.float32, device=mu.device) if out is None else out
    ptr, ld = _row_view(out, n)
    check(_lib.lib().fov_sample_refeed_fwd(_ptr(mu), _ptr(var), _ptr(noise), ptr, ld, B, n // 3, 0 if std == "sqrt" else 1,
                                           1 if planar else 0, _stream()))
    return out


def sample_refeed_bwd(dx, var, noise, dmu, dvar, std="sqrt", planar=False, accumulate=True):
    """Reparameterisation gradient of sample_refeed into dmu / dvar (B,3); dx (B,3*fps) may be row-strided."""
    var, noise = _dev(var, "var"), _dev(noise, "noise")
    B, n = noise.shape
    ptr, ld = _row_view(dx, n)
    check(_lib.lib().fov_sample_refeed_bwd(ptr, ld, _ptr(var), _ptr(noise), _ptr(_dev(dmu, "dmu")), _ptr(_dev(dvar, "dvar")), B,
                                           n // 3, 0 if std == "sqrt" else 1, 1 if planar else 0, 1 if accumulate else 0, _stream()))
    return dmu, dvar


def dense_mse_head_supported(N, H, O):
    return bool(_lib.lib().fov_dense_mse_head_supported(int(N), int(H), int(O)))


def dense_mse_head(hs, W, b, target, activation="tanh", dW=None, db=None, loss=None, weight=1.0, need_dx=True, need_y=True, scratch=None,
                   dx=None, y=None):
    """Dense(O, activation) + Keras mean_squared_error over the rows of hs (..., H), forward AND backward in one launch (rows <= 4096,
    O <= 8: dense_mse_head_supported) -> (y like target or None, dX like hs or None, loss (1,)); dW (H,O), db (O) are written in
    place (they may be views of a flat gradient buffer)."""
    hs, W, b, target = _dev(hs, "hs"), _dev(W, "W"), _dev(b, "b"), _dev(target, "target")
    H, O = W.shape
    N = hs.numel() // H
    assert target.numel() == N * O and dW is not None and db is not None
    if need_y and y is None:
        y = torch.empty(target.shape, dtype=torch.float32, device=hs.device)
    if need_dx and dx is None:
        dx = torch.empty(hs.shape, dtype=torch.float32, device=hs.device)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=hs.device)
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(L.fov_dense_mse_head_workspace_bytes(N, H, O), hs.device)
    check(L.fov_dense_mse_head(_ptr(hs), _ptr(W), _ptr(b), _ptr(target), None if y is None else _ptr(y), None if dx is None else _ptr(dx),
                               _ptr(_dev(dW, "dW")), _ptr(_dev(db, "db")), _ptr(loss), N, H, O, 1 if activation == "tanh" else 0, float(weight),
                               buf.data_ptr(), buf.numel(), _stream()))
    return y, dx, loss


def rmsprop_tf_step(params, grads, ms, lr, decay=0.9, eps=1e-10, clip_value=0.0, guards=None, applied=None):
    """tf.train.RMSPropOptimizer + clip_by_value on flat buffers (lstm.py:556-567); guards / applied as in adam_step."""
    for t in (params, grads, ms):
        _dev(t, "flat buffer")
    if guards is None and applied is None:
        check(_lib.lib().fov_rmsprop_tf_step(_ptr(params), _ptr(grads), _ptr(ms), params.numel(), lr, decay, eps, clip_value,
                                             _stream()))
    else:
        check(_lib.lib().fov_rmsprop_tf_step_guarded(_ptr(params), _ptr(grads), _ptr(ms), params.numel(), lr, decay, eps, clip_value,
                                                     *_guard_ptrs(guards), _applied_ptr(applied), _stream()))


def dense_bwd(x, W, dpre, dW=None, db=None, need_dx=True, accumulate=False, scratch=None, need_dW=True, need_db=True,
              dtype="f32"):
    x, W, dpre = _dev(x, "x"), _dev(W, "W"), _dev(dpre, "dpre")
    In, Out = W.shape
    x2, d2 = x.reshape(-1, In), dpre.reshape(-1, Out)
    N = x2.shape[0]
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
    dW = (e(In, Out) if dW is None else dW) if need_dW else None
    db = (e(Out) if db is None else db) if need_db else None
    dx = e(N, In) if need_dx else None
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(L.fov_dense_bwd_workspace_bytes(N, In, Out), x.device)
    fn = L.fov_dense_bwd_bf16 if dtype == "bf16" else L.fov_dense_bwd
    check(fn(_ptr(x2), _ptr(W), _ptr(d2), _ptr(dx), _ptr(dW), _ptr(db), N, In, Out,
             1 if accumulate else 0, buf.data_ptr(), buf.numel(), _stream()))
    return (dx.reshape(*x.shape[:-1], In) if need_dx else None), dW, db


def mix_head_wgrad(h2, dpre_p, others, p, dpre_m, out, accumulate=False, scratch=None):
    """[dense_W ; dense_b ; mix_W ; mix_b] gradients of the others-mixing head over all steps in one launch + one reduce.
    h2 (T,B,H), dpre_p / p / dpre_m (T,B,O) time-major tapes, others (B,T,U-1,6) or (B,T,n_oth) as the model receives it; out = the
    flat-buffer span of the four tensors ((H + 1 + n_oth + O + 1) * O floats)."""
    h2, dpre_p, others, p, dpre_m, out = (_dev(a, n) for a, n in ((h2, "h2"), (dpre_p, "dpre_p"), (others, "others"), (p, "p"),
                                                                      (dpre_m, "dpre_m"), (out, "out")))
    T, B, H = h2.shape
    O = p.shape[-1]
    n_oth = others.numel() // (B * T)
    assert others.shape[0] == B and others.shape[1] == T and p.shape == (T, B, O) and dpre_p.shape == p.shape and dpre_m.shape == p.shape
    assert out.numel() == (H + 1 + n_oth + O + 1) * O
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(L.fov_mix_head_wgrad_workspace_bytes(B, T, H, O, n_oth), h2.device)
    check(L.fov_mix_head_wgrad(_ptr(h2), _ptr(dpre_p), _ptr(others), _ptr(p), _ptr(dpre_m), _ptr(out), B, T, H, O, n_oth,
                               1 if accumulate else 0, buf.data_ptr(), buf.numel(), _stream()))
    return out


def wgrad_fused(x1, x2, dpre, out, bias=True, accumulate=False, scratch=None, dtype="f32"):
    """out (In1 + In2 + bias, Out) (+)= [x1 | x2 | 1]^T dpre over all rows: a layer's dK, dR and db in ONE product and one
    reduce, written where the three lie adjacent in a flat gradient buffer.  x2 may be None."""
    x1, dpre, out = _dev(x1, "x1"), _dev(dpre, "dpre"), _dev(out, "out")
    Out = dpre.shape[-1]
    In1 = x1.shape[-1]
    N = dpre.numel() // Out
    In2 = 0
    if x2 is not None:
        x2 = _dev(x2, "x2")
        In2 = x2.shape[-1]
        assert x2.numel() == N * In2
    assert x1.numel() == N * In1 and out.numel() == (In1 + In2 + (1 if bias else 0)) * Out
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(L.fov_wgrad_fused_workspace_bytes(N, In1, In2, Out), x1.device)
    check(L.fov_wgrad_fused(_ptr(x1), In1, _ptr(x2), In2, _ptr(dpre), _ptr(out), N, Out, 1 if bias else 0, 1 if accumulate else 0,
                            1 if dtype == "bf16" else 0, buf.data_ptr(), buf.numel(), _stream()))
    return out


def mse_dense_grad(y, target, activation="tanh", scratch=None, dpre=None, loss=None, weight=1.0, time_major=False, db=None):
    """Keras mean_squared_error + Dense activation derivative -> (dpre like y, loss (1,) tensor), both multiplied by
    `weight` (a rank's share n_local / n_global under data parallelism).  time_major: y / dpre are (T,B,O) while
    target is (B,T,O) - the tape layout of the unrolled decoders, no transposed copies.  db (O,): also the head's bias
    gradient (column sums of dpre), from the same launch - then dense_bwd is called with db=None."""
    y, target = _dev(y, "y"), _dev(target, "target")
    n = y.numel()
    tmB = tmT = O = 0
    if time_major:
        tmT, tmB, O = y.shape
        assert target.shape == (tmB, tmT, O)
    else:
        assert y.shape == target.shape
    dpre = torch.empty_like(y) if dpre is None else dpre
    loss = torch.empty(1, dtype=torch.float32, device=y.device) if loss is None else loss
    if db is not None:
        assert not time_major
        O = y.shape[-1]
        assert _dev(db, "db").shape == (O,)
        buf = (scratch or _default_scratch).get(4 * (9 * ((n + 255) // 256) + 64 + 256 * O), y.device)
        check(_lib.lib().fov_mse_dense_grad_db(_ptr(y), _ptr(target), _ptr(dpre), _ptr(loss), _ptr(db), n, O,
                                               1 if activation == "tanh" else 0, float(weight), buf.data_ptr(), buf.numel(), _stream()))
        return dpre, loss
    buf = (scratch or _default_scratch).get(4 * ((n + 255) // 256 + 64), y.device)
    check(_lib.lib().fov_mse_dense_grad_w(_ptr(y), _ptr(target), _ptr(dpre), _ptr(loss), n,
                                          1 if activation == "tanh" else 0, float(weight), tmB, tmT, O,
                                          buf.data_ptr(), buf.numel(), _stream()))
    return dpre, loss


def scale_(x, s):
    """x *= s in place (HIP kernel; torch computes nothing on the product path)."""
    x = _dev(x, "x")
    check(_lib.lib().fov_scale(_ptr(x), x.numel(), float(s), _stream()))
    return x


def _guard_ptrs(guards):
    g = [t.data_ptr() for t in (guards or [])][:3]
    return g + [None] * (3 - len(g))


def guard_flag(guards, out):
    """out (one float of the flat gradient buffer) = 1.0 if a guard workspace's timeout word is set, else 0.0 - the slot
    the data-parallel all-reduce carries so that every rank skips a failed step together."""
    check(_lib.lib().fov_guard_flag(*_guard_ptrs(guards), _ptr(out), _stream()))
    return out


def _applied_ptr(applied):
    if applied is None:
        return None
    assert applied.is_cuda and applied.dtype == torch.int64 and applied.numel() == 1
    return applied.data_ptr()


def adam_step(params, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, guards=None, applied=None):
    """Keras Adam on flat buffers.  guards: up to three workspace buffers; the update is skipped on the device when
    one of their sticky timeout words is set.  applied: (1,) int64 device counter of the updates that were not skipped."""
    for t in (params, grads, m, v):
        _dev(t, "flat buffer")
    check(_lib.lib().fov_adam_step_guarded(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, beta1, beta2,
                                           eps, int(step), *_guard_ptrs(guards), _applied_ptr(applied), _stream()))


def reduce_defer_begin(grad, arena):
    """From here to reduce_defer_end the split weight-gradient products that write into `grad` (a contiguous fp32 view of the
    trainer's flat gradient buffer) leave their partial slices in `arena` and are summed by ONE launch at the flush."""
    check(_lib.lib().fov_reduce_defer_begin(_ptr(grad), grad.numel(), arena.data_ptr(), arena.numel() * arena.element_size(), _stream()))


def reduce_defer_flush(grad=None):
    """Sum the pending products of the region opened on `grad` now (None: of every open region of the process)."""
    check(_lib.lib().fov_reduce_defer_flush(None if grad is None else _ptr(grad), _stream()))


def reduce_defer_end(grad=None):
    """Flush and close the region opened on `grad` (None: every open region of the process)."""
    check(_lib.lib().fov_reduce_defer_end(None if grad is None else _ptr(grad), _stream()))


def rmsprop_step(params, grads, accum, lr=1e-3, rho=0.9, eps=1e-7, guards=None, applied=None):
    for t in (params, grads, accum):
        _dev(t, "flat buffer")
    check(_lib.lib().fov_rmsprop_step_guarded(_ptr(params), _ptr(grads), _ptr(accum), params.numel(), lr, rho, eps,
                                              *_guard_ptrs(guards), _applied_ptr(applied), _stream()))


# ---------------------------------------------------------------------------------------------
# stacked layers / others mixing (a4): building blocks
# ---------------------------------------------------------------------------------------------
def matmul(a, b, scratch=None):
    """(M,K) @ (K,N) on the fp32 MFMA GEMM of the library."""
    a, b = _dev(a, "a"), _dev(b, "b")
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(min(L.fov_matmul_workspace_bytes(M, K, N), 64 << 20), a.device)
    check(L.fov_matmul(_ptr(a), _ptr(b), _ptr(c), M, K, N, buf.data_ptr(), buf.numel(), _stream()))
    return c


def dense_add(x, W, b, add, activation="tanh", out=None):
    """act(x W + b + add) with add (N,Out) possibly a strided row view (last dim contiguous)."""
    x, W = _dev(x, "x"), _dev(W, "W")
    In, Out = W.shape
    x2 = x.reshape(-1, In)
    N = x2.shape[0]
    assert add.is_cuda and add.dtype == torch.float32 and add.shape == (N, Out) and add.stride(1) == 1
    y = torch.empty((N, Out), dtype=torch.float32, device=x.device) if out is None else _dev(out, "out")
    assert y.numel() == N * Out
    check(_lib.lib().fov_dense_add_fwd(_ptr(x2), _ptr(W), _ptr(_dev(b, "b")), add.data_ptr(), add.stride(0), _ptr(y),
                                       N, In, Out, 1 if activation == "tanh" else 0, _stream()))
    return y


def mix_decoder_supported(H, O):
    return H == 256 and O <= 8


def mix_decoder(dec0, h1, c1, h2, c2, oth_proj, w, mix_Wp, T_out, act="sigmoid", workspace=None, out=None, train=None,
                final_state=None, dtype="f32"):
    """The whole unrolled others-mixing decoder in one launch -> out (T_out,B,O) step-major.
    w: dict with dec1_K/R/b, dec2_K/R/b, dense_W/b; oth_proj (B,T_out,O) view (last dim contiguous);
    train: optional dict of preallocated P, H1, C1, H2, C2 (T_out,B,.), res1, res2 (T_out,B,5,H);
    final_state: optional (h1T, c1T, h2T, c2T) tensors."""
    dec0, h1, c1, h2, c2 = (_dev(t, "state") for t in (dec0, h1, c1, h2, c2))
    B, H = h1.shape
    O = w["dense_W"].shape[1]
    assert oth_proj.is_cuda and oth_proj.dtype == torch.float32 and oth_proj.shape == (B, T_out, O) and oth_proj.stride(2) == 1
    out = torch.empty((T_out, B, O), dtype=torch.float32, device=h1.device) if out is None else _dev(out, "out")
    L = _lib.lib()
    ws = workspace or default_workspace(h1.device)
    buf = ws.get(L.fov_mix_decoder_workspace_bytes(B, H), h1.device)
    tr = [None] * 7 if train is None else [_dev(train[k], k) for k in ("P", "H1", "C1", "H2", "C2", "res1", "res2")]
    fs = [None] * 4 if final_state is None else [_dev(t, "final state") for t in final_state]
    fwd = L.fov_mix_decoder_fwd_bf16 if dtype == "bf16" else L.fov_mix_decoder_fwd
    check(fwd(_ptr(dec0.reshape(B, O)), _ptr(h1), _ptr(c1), _ptr(h2), _ptr(c2), oth_proj.data_ptr(),
              oth_proj.stride(0), oth_proj.stride(1),
              _ptr(_dev(w["dec1_K"], "K1")), _ptr(_dev(w["dec1_R"], "R1")), _ptr(_dev(w["dec1_b"], "b1")),
              _ptr(_dev(w["dec2_K"], "K2")), _ptr(_dev(w["dec2_R"], "R2")), _ptr(_dev(w["dec2_b"], "b2")),
              _ptr(_dev(w["dense_W"], "Wd")), _ptr(_dev(w["dense_b"], "bd")), _ptr(_dev(mix_Wp, "Wp")),
              _ptr(out), *[_ptr(t) for t in fs], *[_ptr(t) for t in tr],
              B, T_out, H, O, act_code(act), buf.data_ptr(), buf.numel(), _stream()))
    return out


def mix_decoder_prepack(dec2_K, B, H, workspace=None, workspace_bwd=None):
    """Pack dec2_K for the fused fp32 decoder kernels ahead of their launches (current stream: a side stream under the
    encoder); the next mix_decoder / mix_decoder_bwd on these workspaces skips its own pack."""
    L = _lib.lib()
    dec2_K = _dev(dec2_K, "dec2_K")
    bf = workspace.get(L.fov_mix_decoder_workspace_bytes(B, H), dec2_K.device) if workspace is not None else None
    bb = workspace_bwd.get(L.fov_mix_decoder_bwd_workspace_bytes(B, H), dec2_K.device) if workspace_bwd is not None else None
    check(L.fov_mix_decoder_prepack(_ptr(dec2_K), None if bf is None else bf.data_ptr(), 0 if bf is None else bf.numel(),
                                    None if bb is None else bb.data_ptr(), 0 if bb is None else bb.numel(), H, _stream()))


def mix_decoder_bwd(M, P, dloss, res1, res2, C1, C2, w, mix_Wp, out, act="sigmoid", workspace=None, dtype="f32"):
    """BPTT through the unrolled decoder in one launch.  M, P, dloss (T_out,B,O); res1, res2 (T_out,B,5,H); C1, C2
    (>= T_out rows of (B,H), row t = cell state before step t); out: dict with preallocated DZ1, DZ2 (T_out,B,4H),
    dpre_m, dpre_p (T_out,B,O), dh1_0, dc1_0, dh2_0, dc2_0 (B,H)."""
    M, P, dloss = _dev(M, "M"), _dev(P, "P"), _dev(dloss, "dloss")
    T_out, B, O = M.shape
    H = w["dec1_R"].shape[0]
    L = _lib.lib()
    ws = workspace or Workspace()
    buf = ws.get(L.fov_mix_decoder_bwd_workspace_bytes(B, H), M.device)
    names = ("DZ1", "DZ2", "dpre_m", "dpre_p", "dh1_0", "dc1_0", "dh2_0", "dc2_0")
    bwd = L.fov_mix_decoder_bwd_bf16 if dtype == "bf16" else L.fov_mix_decoder_bwd
    check(bwd(_ptr(M), _ptr(P), _ptr(dloss), _ptr(_dev(res1, "res1")), _ptr(_dev(res2, "res2")),
              _ptr(_dev(C1, "C1")), _ptr(_dev(C2, "C2")),
              _ptr(_dev(w["dec1_K"], "K1")), _ptr(_dev(w["dec1_R"], "R1")),
              _ptr(_dev(w["dec2_K"], "K2")), _ptr(_dev(w["dec2_R"], "R2")),
              _ptr(_dev(w["dense_W"], "Wd")), _ptr(_dev(mix_Wp, "Wp")),
              *[_ptr(_dev(out[k], k)) for k in names],
              B, T_out, H, O, act_code(act), buf.data_ptr(), buf.numel(), _stream()))
    return out


def mix_head_fwd(h, dense_W, dense_b, mix_Wp, add, p_out, m_out):
    """p = tanh(h dense_W + dense_b), m = tanh(p mix_Wp + add) in one launch; add (N,O) may be a strided row view."""
    h, dense_W, dense_b, mix_Wp = _dev(h, "h"), _dev(dense_W, "dense_W"), _dev(dense_b, "dense_b"), _dev(mix_Wp, "mix_Wp")
    N, H = h.shape
    O = dense_W.shape[1]
    assert add.is_cuda and add.shape == (N, O) and add.stride(1) == 1 and mix_Wp.shape == (O, O)
    check(_lib.lib().fov_mix_head_fwd(_ptr(h), _ptr(dense_W), _ptr(dense_b), _ptr(mix_Wp), add.data_ptr(), add.stride(0),
                                      _ptr(_dev(p_out, "p")), _ptr(_dev(m_out, "m")), N, H, O, _stream()))
    return p_out, m_out


def mix_head_bwd(dm_loss, dm_feedback, m, p, mix_Wp, dense_W, dpre_m, dpre_p, dh=None):
    """Backward of mix_head_fwd -> dh (N,H); dpre_m / dpre_p (N,O) are written (no aliasing with the inputs)."""
    dm_loss, m, p = _dev(dm_loss, "dm_loss"), _dev(m, "m"), _dev(p, "p")
    N, O = m.shape
    H = dense_W.shape[0]
    dh = torch.empty((N, H), dtype=torch.float32, device=m.device) if dh is None else dh
    check(_lib.lib().fov_mix_head_bwd(_ptr(dm_loss), _ptr(_dev(dm_feedback, "dm_feedback")), _ptr(m), _ptr(p),
                                      _ptr(_dev(mix_Wp, "mix_Wp")), _ptr(_dev(dense_W, "dense_W")), _ptr(_dev(dpre_m, "dpre_m")),
                                      _ptr(_dev(dpre_p, "dpre_p")), _ptr(dh), N, H, O, _stream()))
    return dh


def lstm_seq_zx(zx, R, b, h0=None, c0=None, act="sigmoid", impl="auto", return_sequences=True, workspace=None,
                reserve=None, out=None):
    """LSTM layer from a precomputed input projection zx = x.K (B,T,4H) -> (hs|None, hT, cT).
    `out` may carry preallocated (hs|None, hT|None, cT|None) tensors (None entries are not written)."""
    zx, R, b = _dev(zx, "zx"), _dev(R, "R"), _dev(b, "b")
    B, T, H4 = zx.shape
    H = R.shape[0]
    assert H4 == 4 * H
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=zx.device)
    if out is None:
        hs = e(B, T, H) if return_sequences else None
        hT, cT = e(B, H), e(B, H)
    else:
        hs, hT, cT = (_dev(t, "out") for t in out)
    L = _lib.lib()
    impl = impl_code(impl)
    ws = (workspace or default_workspace(zx.device))
    buf = ws.get(L.fov_lstm_seq_workspace_bytes(B, T, 1, H, impl), zx.device)
    check(L.fov_lstm_seq_fwd_zx(_ptr(zx), _ptr(R), _ptr(b), _ptr(_dev(h0, "h0")), _ptr(_dev(c0, "c0")), _ptr(hs),
                                _ptr(hT), _ptr(cT), _ptr(reserve), B, T, H, act_code(act), impl, buf.data_ptr(),
                                buf.numel(), _stream()))
    return hs, hT, cT


# ---------------------------------------------------------------------------------------------
# ConvLSTM2D seq2seq (a8/a9): building blocks
# ---------------------------------------------------------------------------------------------
def conv2d(x, w, b=None, add=None, activation=None, out=None, in_channels=None, dilation=1):
    """y (B,H,W,N) = act(conv2d_same(x, w) + b + add).  x may be a channel slice view of a wider NHWC map
    (only the last-dim stride may differ from dense: pass the view, pixel stride is taken from it).
    dilation = Keras `dilation_rate` (taps `dilation` pixels apart, 'same' padding grown to match)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.stride(3) == 1
    B, H, W, C = x.shape
    ldx = x.stride(2)
    ldb = x.stride(0) if B > 1 else H * W * ldx
    assert x.stride(1) == W * ldx and ldb >= H * W * ldx, "x must be NHWC with a uniform pixel stride"
    w = _dev(w, "w")
    kh, kw, Cw, N = w.shape
    assert Cw == C
    y = torch.empty((B, H, W, N), dtype=torch.float32, device=x.device) if out is None else out
    act = {None: 0, "linear": 0, "relu": 2}[activation]
    if dilation != 1:
        check(_lib.lib().fov_conv2d_dilated_fwd(x.data_ptr(), ldx, ldb, _ptr(w), _ptr(_dev(b, "b")), _ptr(add), _ptr(y), B, H, W,
                                                C, N, kh, kw, int(dilation), act, _stream()))
        return y
    check(_lib.lib().fov_conv2d_fwd(x.data_ptr(), ldx, ldb, _ptr(w), _ptr(_dev(b, "b")), _ptr(add), _ptr(y), B, H, W, C, N,
                                    kh, kw, act, _stream()))
    return y


def conv2d_cat(x1, x2, w, b=None, activation=None, out=None):
    """y = act(conv2d_same([x1 | x2], w) + b): convolution over the channel concatenation of two NHWC maps (each may
    be a channel-slice / batch-strided view) without materialising it; w (kh,kw,C1+C2,N)."""
    def geom(x):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.stride(3) == 1
        B, H, W, C = x.shape
        ldx = x.stride(2)
        ldb = x.stride(0) if B > 1 else H * W * ldx
        assert x.stride(1) == W * ldx and ldb >= H * W * ldx, "NHWC with a uniform pixel stride"
        return B, H, W, C, ldx, ldb
    B, H, W, C1, ldx1, ldb1 = geom(x1)
    B2, H2, W2, C2, ldx2, ldb2 = geom(x2)
    assert (B, H, W) == (B2, H2, W2)
    w = _dev(w, "w")
    kh, kw, Cw, N = w.shape
    assert Cw == C1 + C2
    y = torch.empty((B, H, W, N), dtype=torch.float32, device=x1.device) if out is None else out
    act = {None: 0, "linear": 0, "relu": 2}[activation]
    check(_lib.lib().fov_conv2d_fwd2(x1.data_ptr(), ldx1, ldb1, C1, x2.data_ptr(), ldx2, ldb2, C2, _ptr(w), _ptr(_dev(b, "b")),
                                     None, _ptr(y), B, H, W, N, kh, kw, act, _stream()))
    return y


def convlstm_cell(x, h_prev, w, b, c_prev, h_out, act="hard_sigmoid", c_new=None, gates=None, dilation=1):
    """One ConvLSTM2D step in one launch: -> (h_out, c_new).  x (B,H,W,C) and h_prev (B,H,W,F) NHWC maps (channel-slice /
    batch-strided views allowed; h_prev None = zero state, w is then K alone); w (kh,kw,C+F,4F) = [K ; R]; c_prev None = zero
    state; c_new defaults to updating c_prev in place; h_out may be a channel-slice view and must not be h_prev; gates
    (B,H,W,4F) receives the activated i,f,g,o when given (training tape).  dilation spreads the taps over x only, as Keras's
    ConvLSTM2D does (its recurrent convolution is never dilated)."""
    def geom(t):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.stride(3) == 1
        B, H, W, C = t.shape
        ldx = t.stride(2)
        ldb = t.stride(0) if B > 1 else H * W * ldx
        assert t.stride(1) == W * ldx and ldb >= H * W * ldx, "NHWC with a uniform pixel stride"
        return B, H, W, C, ldx, ldb
    B, H, W, C, ldx, ldb = geom(x)
    w = _dev(w, "w")
    kh, kw, Cw, N = w.shape
    F = N // 4
    assert N == 4 * F
    ldx2 = ldb2 = 0
    if h_prev is not None:
        B2, H2, W2, F2, ldx2, ldb2 = geom(h_prev)
        assert (B2, H2, W2, F2) == (B, H, W, F) and Cw == C + F
    else:
        assert Cw == C
    c_prev = _dev(c_prev, "c_prev")
    if c_new is None:
        c_new = c_prev if c_prev is not None else torch.empty((B, H, W, F), dtype=torch.float32, device=x.device)
    c_new = _dev(c_new, "c_new")
    assert c_new.shape == (B, H, W, F) and (c_prev is None or c_prev.shape == c_new.shape)
    assert h_out.is_cuda and h_out.dtype == torch.float32 and h_out.stride(-1) == 1 and h_out.shape == c_new.shape
    assert h_out.stride(1) == W * h_out.stride(2) and (B == 1 or h_out.stride(0) == H * W * h_out.stride(2)), "h_out: uniform pixel stride"
    if gates is not None:
        gates = _dev(gates, "gates")
        assert gates.shape == (B, H, W, N)
    _sync_env()
    if dilation != 1:
        check(_lib.lib().fov_convlstm_cell_dilated_fwd(x.data_ptr(), ldx, ldb, C, h_prev.data_ptr() if h_prev is not None else None,
                                                       ldx2, ldb2, _ptr(w), _ptr(_dev(b, "b")), _ptr(c_prev), _ptr(c_new),
                                                       h_out.data_ptr(), h_out.stride(-2), _ptr(gates), B, H, W, F, kh, kw,
                                                       int(dilation), act_code(act), _stream()))
        return h_out, c_new
    check(_lib.lib().fov_convlstm_cell_fwd(x.data_ptr(), ldx, ldb, C, h_prev.data_ptr() if h_prev is not None else None, ldx2, ldb2,
                                           _ptr(w), _ptr(_dev(b, "b")), _ptr(c_prev), _ptr(c_new), h_out.data_ptr(),
                                           h_out.stride(-2), _ptr(gates), B, H, W, F, kh, kw, act_code(act), _stream()))
    return h_out, c_new


def convlstm_gates(z, c, h_out, act="hard_sigmoid"):
    """Gates + cell update: z (B,H,W,4F), c (B,H,W,F) updated in place, h written into h_out, which may be
    a channel-slice view of a concatenated feature map."""
    z, c = _dev(z, "z"), _dev(c, "c")
    F = c.shape[-1]
    rows = c.numel() // F
    assert h_out.is_cuda and h_out.stride(-1) == 1 and h_out.shape == c.shape
    check(_lib.lib().fov_convlstm_gates(_ptr(z), _ptr(c), h_out.data_ptr(), h_out.stride(-2), rows, F, act_code(act),
                                        _stream()))
    return h_out, c


def softmax_lastdim(x, out=None):
    x = _dev(x, "x")
    y = torch.empty_like(x) if out is None else _dev(out, "out")
    n = x.shape[-1]
    check(_lib.lib().fov_softmax_lastdim(_ptr(x), _ptr(y), x.numel() // n, n, _stream()))
    return y


def convlstm_gates_train(z, c_prev, h_out, act="hard_sigmoid", gates=None, c_new=None):
    """Training forward of the gates: -> (h_out, c_new, gates).  gates (B,H,W,4F) keeps the activated i,f,g,o
    (defaults to overwriting z); c_prev None = zero state; h_out may be a channel-slice view."""
    z = _dev(z, "z")
    F = z.shape[-1] // 4
    rows = z.numel() // (4 * F)
    gates = z if gates is None else gates
    c_new = torch.empty(z.shape[:-1] + (F,), dtype=torch.float32, device=z.device) if c_new is None else _dev(c_new, "c_new")
    assert h_out.is_cuda and h_out.stride(-1) == 1 and h_out.shape == c_new.shape
    check(_lib.lib().fov_convlstm_gates_train(_ptr(z), _ptr(_dev(c_prev, "c_prev")), _ptr(c_new), h_out.data_ptr(),
                                              h_out.stride(-2), _ptr(gates), rows, F, act_code(act), _stream()))
    return h_out, c_new, gates


def convlstm_gates_bwd(dh, dc, gates, c_prev, c_new, act="hard_sigmoid", dz=None):
    """dh (B,H,W,F) (may be a channel-slice view), dc updated in place (dL/dc_t -> dL/dc_{t-1}) -> dz (B,H,W,4F)."""
    gates, c_new, dc = _dev(gates, "gates"), _dev(c_new, "c_new"), _dev(dc, "dc")
    F = c_new.shape[-1]
    rows = c_new.numel() // F
    assert dh.is_cuda and dh.dtype == torch.float32 and dh.stride(-1) == 1 and dh.shape == c_new.shape
    dz = torch.empty_like(gates) if dz is None else dz
    check(_lib.lib().fov_convlstm_gates_bwd(dh.data_ptr(), dh.stride(-2), _ptr(dc), _ptr(gates), _ptr(_dev(c_prev, "c_prev")),
                                            _ptr(c_new), _ptr(dz), rows, F, act_code(act), _stream()))
    return dz


def conv2d_wgrad(x, dy, kh, kw, dw=None, accumulate=False, scratch=None, dilation=1):
    """dw (kh,kw,C,N) (+)= weight gradient of y = conv2d_same(x, w).  x: batch-dense NHWC (leading dims are
    flattened into the batch; the last dim may be a channel slice of a wider map); dy (..., N) dense."""
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1
    H, W, C = x.shape[-3:]
    ldx = x.stride(-2)
    B = x.numel() // (H * W * C)
    expect = W * ldx     # stride of the H axis, then of every leading (batch / time) axis; size-1 axes are free
    for i in range(x.dim() - 3, -1, -1):
        assert x.shape[i] == 1 or x.stride(i) == expect, "x must be batch-dense NHWC"
        expect *= x.shape[i]
    dy = _dev(dy, "dy")
    N = dy.shape[-1]
    assert dy.numel() == B * H * W * N
    dw = torch.empty((kh, kw, C, N), dtype=torch.float32, device=x.device) if dw is None else dw
    L = _lib.lib()
    buf = (scratch or _default_scratch).get(L.fov_conv2d_wgrad_workspace_bytes(C, N, kh, kw), x.device)
    if dilation != 1:
        check(L.fov_conv2d_dilated_wgrad(x.data_ptr(), ldx, _ptr(dy), _ptr(dw), B, H, W, C, N, kh, kw, int(dilation),
                                         1 if accumulate else 0, buf.data_ptr(), buf.numel(), _stream()))
        return dw
    check(L.fov_conv2d_wgrad(x.data_ptr(), ldx, _ptr(dy), _ptr(dw), B, H, W, C, N, kh, kw, 1 if accumulate else 0,
                             buf.data_ptr(), buf.numel(), _stream()))
    return dw


def conv2d_weight_transpose(w, out=None):
    """(kh,kw,C,N) -> (kh,kw,N,C) flipped spatially: dx = conv2d(dy, conv2d_weight_transpose(w))."""
    w = _dev(w, "w")
    kh, kw, C, N = w.shape
    out = torch.empty((kh, kw, N, C), dtype=torch.float32, device=w.device) if out is None else out
    check(_lib.lib().fov_conv2d_weight_transpose(_ptr(w), _ptr(out), kh, kw, C, N, _stream()))
    return out


def softmax_lastdim_bwd(dp, p, out=None):
    dp, p = _dev(dp, "dp"), _dev(p, "p")
    n = p.shape[-1]
    out = torch.empty_like(p) if out is None else out
    check(_lib.lib().fov_softmax_lastdim_bwd(_ptr(dp), _ptr(p), _ptr(out), p.numel() // n, n, _stream()))
    return out


def colsum(x, out=None, accumulate=False, scratch=None):
    """Column sums of x (..., cols) over all leading dims (bias gradients)."""
    x = _dev(x, "x")
    cols = x.shape[-1]
    out = torch.empty(cols, dtype=torch.float32, device=x.device) if out is None else out
    buf = (scratch or _default_scratch).get(4 * (256 * cols + 64), x.device)
    check(_lib.lib().fov_colsum(_ptr(x), _ptr(out), x.numel() // cols, cols, 1 if accumulate else 0, buf.data_ptr(),
                                buf.numel(), _stream()))
    return out


def fov_hit_rate(pred, gt_xyz, span_deg=120.0, gt_span_deg=120.0):
    """Hit rate per (sequence, second).  pred: (..., >=3) device tensor whose first three columns are the
    predicted mean x,y,z (e.g. the model's (N,T,6) output); gt_xyz: (..., >=3) likewise (e.g. meanvar_xyz of the
    ground-truth second).  -> (...,) float32."""
    assert pred.is_cuda and gt_xyz.is_cuda and pred.dtype == torch.float32 and gt_xyz.dtype == torch.float32
    assert pred.shape[:-1] == gt_xyz.shape[:-1] and pred.is_contiguous() and gt_xyz.is_contiguous()
    lead = pred.shape[:-1]
    rows = 1
    for d in lead:
        rows *= d
    out = torch.empty(lead, dtype=torch.float32, device=pred.device)
    check(_lib.lib().fov_fov_hit_rate(pred.data_ptr(), pred.shape[-1], gt_xyz.data_ptr(), gt_xyz.shape[-1], _ptr(out), rows,
                                      float(span_deg), float(gt_span_deg), _stream()))
    return out


def window_stacks(x, T, stride, collapse_user=True):
    """reshape2second_stacks on the device: x (U,S,feat) -> (enc, future, future_input)."""
    x = _dev(x, "x")
    U, S, feat = x.shape
    L = _lib.lib()
    W = int(L.fov_window_count(S, T, stride))
    shape = (W * U, T, feat) if collapse_user else (U, W, T, feat)
    outs = [torch.empty(shape, dtype=torch.float32, device=x.device) for _ in range(3)]
    check(L.fov_window_stacks(_ptr(x), *[_ptr(o) for o in outs], U, S, feat, T, stride, 1 if collapse_user else 0, _stream()))
    return tuple(outs)
