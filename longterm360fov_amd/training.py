"""One training step of the teacher-forced seq2seq graph on the GPU (model.fit's inner loop,
mycode/FoV_seq2seq.py:103,112-117): forward with reserve -> MSE -> Dense backward -> decoder BPTT
-> encoder BPTT -> (data-parallel: ONE all-reduce of the flat gradient buffer) -> Keras Adam /
RMSprop on the flat parameter buffer.  All arithmetic happens in libfov360_hip.so; torch holds
the buffers and runs the RCCL collective."""
import os

import numpy as np
import torch

from . import ops, parallel

_W_ORDER = ("enc_K", "enc_R", "enc_b", "dec_K", "dec_R", "dec_b", "dense_W", "dense_b")


class FlatParamTrainer:
    """What every trainer of the Keras-optimizer models shares: parameters and gradients live in ONE flat buffer each
    (views per tensor in self.w / self.g), so the optimizer is one launch.  The gradient buffer carries one extra
    slot for the loss, so data parallelism needs no collective of its own for it.  A subclass provides
    forward_backward(*inputs, grad_weight=1.0) -> (loss (1,), prediction) filling self.grad (already multiplied by
    grad_weight) and, where it can, calls grads_final(name) as soon as every gradient from parameter `name` to the end
    of the buffer is final: under DP that tail is all-reduced (async, RCCL's own stream) while the rest of the
    backward pass is still running, the head of the buffer follows at the end of the step."""

    def _alloc(self, weights, order, optimizer, lr, device):
        self.optimizer, self.lr, self.device = optimizer.lower(), float(lr), device
        self.shapes = [(k, tuple(weights[k].shape)) for k in order]
        n = int(sum(np.prod(s) for _, s in self.shapes))
        self.flat = torch.empty(n, dtype=torch.float32, device=device)     # parameters, one buffer
        # gradients in the same layout, between a 16-byte head whose first float is the POISON slot (data parallelism: 1.0 if
        # a persistent kernel of this rank's step gave up; the all-reduce makes it every rank's guard) and the loss slot
        self.gradbuf = torch.zeros(4 + n + 1, dtype=torch.float32, device=device)
        self.poison_slot = self.gradbuf[:1]
        self.grad = self.gradbuf[4:4 + n]
        self.loss_slot = self.gradbuf[4 + n:]
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device) if self.optimizer == "adam" else None
        self.w, self.g, self.offset = {}, {}, {}
        off = 0
        for k, s in self.shapes:
            cnt = int(np.prod(s))
            self.w[k] = self.flat[off:off + cnt].view(*s)
            self.g[k] = self.grad[off:off + cnt].view(*s)
            self.w[k].copy_(torch.from_numpy(np.ascontiguousarray(weights[k], dtype=np.float32)))
            self.offset[k] = off
            off += cnt
        self.step_count = 0
        self.applied = torch.zeros(1, dtype=torch.int64, device=device)     # optimizer launches that were not skipped on the device
        self._dp_steps_since_check = 0
        self._dp_seen = False
        self._dp_guard = None
        self.ws = ops.Workspace()
        self.scratch = ops.Scratch()        # split-K partials of the Dense / MSE / matmul calls
        self.bwd_scratch = ops.Scratch()    # BPTT calls only: its head is the persistent kernel's header + granule area
        self._dp = False
        self._pending, self._reduced_from = [], self.gradbuf.numel()

    def _span(self, first, last):
        """The slice of the flat gradient buffer from parameter `first` through parameter `last` (adjacent in the order)."""
        hi = self.offset[last] + self.g[last].numel()
        return self.grad[self.offset[first]:hi]

    def weights_numpy(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.w.items()}

    def load_weights_(self, weights):
        """Overwrite the parameters in place (Keras set_weights / load_weights keep the optimizer state)."""
        for k, _ in self.shapes:
            self.w[k].copy_(torch.from_numpy(np.ascontiguousarray(weights[k], dtype=np.float32)))

    def check(self):
        """Raise FovError(ERR_TIMEOUT) if a persistent kernel of this trainer gave up a bounded wait (synchronises).

        Under data parallelism this is a collective in the sense that every rank must call it at the same points (fit does,
        once per epoch): the timeout word is sticky on the rank that failed, so from the failing step on the all-reduced poison
        slot is nonzero on EVERY rank at every step - all replicas skip those updates together - and one scalar all-reduce
        here tells every rank that some rank failed, in a step or in a forward pass since.  All of them raise, and all set their step counter back
        to the number of updates that really ran (self.applied, counted on the device by the guarded optimizer: the steps
        before the failing one stay counted, Adam's bias correction continues where the parameters are)."""
        stepped, self._dp_steps_since_check = self._dp_steps_since_check > 0, 0
        err = None
        try:
            self.ws.check()
            self.bwd_scratch.check()
            ws_bwd = getattr(self, "ws_bwd", None)
            if ws_bwd is not None:
                ws_bwd.check()
        except Exception as exc:
            err = exc
        # Every rank learns here whether ANY rank's persistent kernel gave up since the last check - in a training step (the
        # all-reduced poison slot said so already and the replicas skipped those updates together) or in a forward that no
        # all-reduce follows (model.fit's validation passes run between the last step and this check): one scalar all-reduce per
        # check, i.e. per epoch, so that all ranks raise together instead of one raising while its peers walk into the next
        # collective.
        peer_failed = False
        if self._dp_seen and parallel.dp_active():
            flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float32, device=self.gradbuf.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.SUM)
            peer_failed = float(flag.item()) > (1.0 if err is not None else 0.0)
        elif stepped:      # (a group of one under FOV_FORCE_DIST, or the group is gone: what the last step's slot says)
            peer_failed = err is None and float(self.poison_slot.item()) != 0.0
        if err is not None:
            self.step_count = int(self.applied.item())
            raise err
        if peer_failed:
            self.step_count = int(self.applied.item())
            from ._lib import ERR_TIMEOUT, FovError
            raise FovError(ERR_TIMEOUT, "a persistent kernel of another data-parallel rank gave up a bounded wait: every rank skipped "
                                        "the optimizer updates from that step on")

    def _guards(self):
        """The workspaces whose sticky timeout word the optimizer launch looks at: if a persistent kernel of this step
        gave up, the update is skipped ON THE DEVICE (no host synchronisation), the parameters stay as they were and
        the next check() reports the failure."""
        if self._dp_guard is not None:   # data parallelism: the all-reduced poison slot stands for every rank's workspaces
            return [self._dp_guard]
        return [b.buf for b in (self.ws, self.bwd_scratch, getattr(self, "ws_bwd", None)) if b is not None and b.buf is not None]

    def apply_gradients(self):
        """Keras Adam / RMSprop on the flat buffer (model.compile(optimizer=...), FoV_seq2seq.py:103, convlstm_seq2seq.py:287)."""
        self.step_count += 1
        if self.optimizer == "adam":
            ops.adam_step(self.flat, self.grad, self.m, self.v, self.step_count, lr=self.lr, guards=self._guards(), applied=self.applied)
        else:
            ops.rmsprop_step(self.flat, self.grad, self.m, lr=self.lr, guards=self._guards(), applied=self.applied)

    # ---- data parallelism: the flat buffer (poison slot + gradients + loss slot) is SUM all-reduced ----
    # One all-reduce when the step's kernels have all been issued, or - opt-in, FOV_DP_OVERLAP=1 / overlap_allreduce - the
    # tail first, under the encoder's BPTT.  Opt-in because the BPTT kernels are persistent, one workgroup on EVERY CU
    # with the whole register file: an RCCL kernel that got CUs first and waits there for a lagging peer keeps some of
    # their workgroups from starting while the others spin for them, and after about a second that is a (collective,
    # see train_step) skipped step.  On ONE GPU RCCL launches no device code for an all-reduce at all (rocprofv3 trace
    # of tools/prof_nccl_step.py, profiles/r03_nccl_world1_kernel_stats.txt), so which of the two orders the hardware
    # picks at 8 GPUs is unmeasured.
    overlap_allreduce = os.environ.get("FOV_DP_OVERLAP", "") == "1"

    def grads_final(self, name):
        """Called from forward_backward: every gradient from parameter `name` to the end of the buffer (and the loss
        slot) is final.  Under DP with overlap_allreduce the tail goes out now, overlapping the backward work still to come."""
        if self._dp and self.overlap_allreduce and 4 + self.offset[name] < self._reduced_from:
            ops.reduce_defer_flush(self.grad)     # the tail is read now
            lo = 4 + self.offset[name]   # (gradbuf index: the gradients sit behind the 16-byte head)
            self._pending.append(torch.distributed.all_reduce(self.gradbuf[lo:self._reduced_from],
                                                              op=torch.distributed.ReduceOp.SUM, async_op=True))
            self._reduced_from = lo

    # One reduce launch per step instead of one per split product (ops.reduce_defer_begin): only for trainers whose
    # forward_backward writes gradients through the library's weight-gradient entry points alone (they keep in-stream
    # order by themselves); the arena holds the partial slices of a step (FOV_DEFER_ARENA_MB, 0 switches the deferral off).
    defer_reduces = False
    _defer_arena = None

    def _defer_begin(self):
        if not self.defer_reduces:
            return False
        if self._defer_arena is None:
            mb = int(os.environ.get("FOV_DEFER_ARENA_MB", "256"))
            self._defer_arena = torch.empty(max(mb, 0) << 18, dtype=torch.float32, device=self.device) if mb > 0 else False
        if self._defer_arena is False:
            return False
        ops.reduce_defer_begin(self.grad, self._defer_arena)
        return True

    def _weigh(self, loss, grad_weight):
        """Fallback for trainers whose loss kernel takes no weight: scale gradients and loss after the fact."""
        if grad_weight != 1.0:
            ops.reduce_defer_flush(self.grad)
            ops.scale_(self.grad, grad_weight)
            loss = ops.scale_(loss, grad_weight)
        return loss

    def train_step(self, *inputs, n_global=None, **kw):
        """One optimizer step on (*model inputs, target).  Under data parallelism every rank passes its shard of the
        global batch and `n_global` = global batch size: gradients and loss are combined as sum_r (n_r/n) x_r by SUM
        all-reduces of the flat buffer (one for the tail named by grads_final, overlapped with the remaining backward
        work, one for the head; a trainer that names no tail does a single one), so the update equals the
        single-process one.  Returns the (global) loss as a (1,) tensor that the NEXT step overwrites."""
        weight = self._begin_step(inputs[0].shape[0], n_global)
        deferring = self._defer_begin()
        try:
            loss, _ = self.forward_backward(*inputs, grad_weight=weight, **kw)
        finally:
            if deferring:
                ops.reduce_defer_end(self.grad)      # flushes: every gradient is final from here on
        return self._finish_step(loss)

    def _begin_step(self, n_local, n_global=None):
        """-> this rank's share n_local / n_global of the global batch (1.0 on one rank): the weight of its loss and gradients."""
        _, world = parallel.world()
        weight = 1.0 if world == 1 else n_local / float(n_global if n_global else n_local * world)
        self._dp = parallel.dp_active()   # more than one rank - or FOV_FORCE_DIST=1 on an initialised group of one (RCCL tests)
        self._pending, self._reduced_from = [], self.gradbuf.numel()
        return weight

    def _finish_step(self, loss):
        """Every gradient of the step is in self.grad: the data-parallel all-reduce of the flat buffer, then the (guarded) optimizer."""
        if self._dp:
            if loss.data_ptr() != self.loss_slot.data_ptr():
                self.loss_slot.copy_(loss.reshape(1))
            # the head goes out last, when every kernel of the step has been issued: its first float says whether one of THIS
            # rank's persistent kernels gave up; summed over ranks it is the guard of every rank's optimizer launch, so the
            # replicas skip a failed step together instead of diverging (a rank that alone skipped would hang the others in
            # the next collective once its check() raises)
            ops.guard_flag(self._guards(), self.poison_slot)
            if self._reduced_from > 0:
                self._pending.append(torch.distributed.all_reduce(self.gradbuf[:self._reduced_from],
                                                                  op=torch.distributed.ReduceOp.SUM, async_op=True))
            for work in self._pending:
                work.wait()      # the launch stream waits for the collective; the host does not block
            self._pending, self._dp = [], False
            loss = self.loss_slot
            self._dp_guard = self.poison_slot
            self._dp_steps_since_check += 1
            self._dp_seen = True
        try:
            self.apply_gradients()
        finally:
            self._dp_guard = None
        return loss

    def eval_loss(self, *inputs):
        """Validation loss through the training forward (gradients are overwritten by the next step)."""
        loss, _ = self.forward_backward(*inputs)
        return loss


def _pad_gates(a, H, Hp):
    """(..., 4H) -> (..., 4Hp): every gate block padded with zero columns."""
    out = np.zeros(a.shape[:-1] + (4 * Hp,), np.float32)
    for g in range(4):
        out[..., g * Hp:g * Hp + H] = a[..., g * H:(g + 1) * H]
    return out


def _unpad_gates(a, H, Hp):
    return np.concatenate([a[..., g * Hp:g * Hp + H] for g in range(4)], axis=-1)


class PaddedTrainer:
    """A trainer run at the next matrix-core width Hp with zero-padded weights, so that `fit` at the widths the reference
    ships (latent_dim = 32 in given_others_gt_mean_var_seq2seq.py:38, 64 in FoV_seq2seq.py:22) takes the persistent MFMA
    kernels instead of the generic VALU one.  Exact, not approximate: a padded unit has z = 0 for all four gates, so
    i = f = o = s(0), g = tanh(0) = 0, c stays 0 and h = 0; its dz is 0 because nothing downstream reads it (zero R
    rows, zero Dense rows), hence every gradient in a padded slice is exactly 0, Adam / RMSprop leave the zeros where
    they are (checked: padded_slices_are_zero), and the real units see exactly the unpadded arithmetic.
    `hidden_inputs`: names of the K tensors whose INPUT is a hidden sequence (stacked layers): their rows are padded too."""

    def __init__(self, make_inner, weights, H, Hp, hidden_inputs=()):
        self.H, self.Hp, self.hidden_inputs = int(H), int(Hp), tuple(hidden_inputs)
        self._names = list(weights)
        self.inner = make_inner(self.pad(weights))

    def _kind(self, k, v):
        H = self.H
        if k.endswith("_R") and v.shape == (H, 4 * H):
            return "R"
        if k.endswith("_K") and v.ndim == 2 and v.shape[1] == 4 * H:
            return "KH" if k in self.hidden_inputs else "K"
        if k.endswith("_b") and v.shape == (4 * H,) and (k[:-2] + "_R") in self._names:
            return "b"
        if k == "dense_W" and v.shape[0] == H:
            return "rows"
        return None

    def pad(self, w):
        H, Hp, out = self.H, self.Hp, {}
        for k in self._names:
            v = np.asarray(w[k], np.float32)
            kind = self._kind(k, v)
            if kind in ("R", "KH"):
                t = np.zeros((Hp, 4 * H), np.float32)
                t[:H] = v
                out[k] = _pad_gates(t, H, Hp)
            elif kind in ("K", "b"):
                out[k] = _pad_gates(v, H, Hp)
            elif kind == "rows":
                t = np.zeros((Hp,) + v.shape[1:], np.float32)
                t[:H] = v
                out[k] = t
            else:
                out[k] = v
        return out

    def unpad(self, wp):
        H, Hp, out = self.H, self.Hp, {}
        for k in self._names:
            v = wp[k]
            ref_shape_kind = None
            if k.endswith("_R") and v.shape == (Hp, 4 * Hp):
                ref_shape_kind = "R"
            elif k.endswith("_K") and v.ndim == 2 and v.shape[1] == 4 * Hp:
                ref_shape_kind = "KH" if k in self.hidden_inputs else "K"
            elif k.endswith("_b") and v.shape == (4 * Hp,) and (k[:-2] + "_R") in self._names:
                ref_shape_kind = "b"
            elif k == "dense_W" and v.shape[0] == Hp:
                ref_shape_kind = "rows"
            if ref_shape_kind in ("R", "KH"):
                out[k] = _unpad_gates(v[:H], H, Hp)
            elif ref_shape_kind in ("K", "b"):
                out[k] = _unpad_gates(v, H, Hp)
            elif ref_shape_kind == "rows":
                out[k] = v[:H].copy()
            else:
                out[k] = v
        return out

    def padded_slices_are_zero(self):
        """True if every padded slice of parameters AND of the last gradients is exactly zero (synchronises)."""
        wp = self.inner.weights_numpy()
        gp = {k: v.detach().cpu().numpy() for k, v in self.inner.g.items()}
        for d in (wp, gp):
            back = self.pad(self.unpad(d))
            for k in self._names:
                if not np.array_equal(back[k], d[k]):
                    return False
        return True

    def weights_numpy(self):
        return self.unpad(self.inner.weights_numpy())

    def load_weights_(self, weights):
        self.inner.load_weights_(self.pad(weights))

    @property
    def lr(self):
        return self.inner.lr

    @lr.setter
    def lr(self, v):
        self.inner.lr = v

    _OWN = ("H", "Hp", "hidden_inputs", "_names", "inner")

    def __getattr__(self, name):      # train_step, eval_loss, forward_backward, check, step_count, ws, ...
        inner = self.__dict__.get("inner")
        if inner is None:             # not constructed yet (copy / pickle probe attributes first)
            raise AttributeError(name)
        return getattr(inner, name)

    def __setattr__(self, name, value):      # overlap_allreduce, defer_reduces, fused_decoder, ...: the inner trainer reads them
        if name in self._OWN or "inner" not in self.__dict__ or name in type(self).__dict__:
            object.__setattr__(self, name, value)
        else:
            setattr(self.__dict__["inner"], name, value)


class Seq2SeqTrainer(FlatParamTrainer):
    defer_reduces = True   # gradients are written by the library's weight-gradient entry points only

    def __init__(self, weights, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda"):
        self.act, self.impl = act, impl
        self._alloc(weights, _W_ORDER, optimizer, lr, device)
        self._bufs = {}

    def _buffers(self, B, T_in, T_out):
        key = (B, T_in, T_out)
        if key not in self._bufs:
            H = self.w["enc_R"].shape[0]
            O = self.w["dense_W"].shape[1]
            e = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
            self._bufs[key] = {
                "enc": (e(B, T_in, H), e(B, H), e(B, H), e(B, T_in, 5, H)),
                "dec": (e(B, T_out, H), e(B, H), e(B, H), e(B, T_out, 5, H)),
                "dz_enc": e(B, T_in, 4 * H), "dz_dec": e(B, T_out, 4 * H),
                "dpre": e(B, T_out, O), "y": e(B, T_out, O), "d_hs": e(B, T_out, H),
            }
        return self._bufs[key]

    def forward_backward(self, enc, dec_in, target, grad_weight=1.0):
        """Fills self.grad with d(mean squared error)/d(parameters) for this (local) batch, scaled by
        `grad_weight`; returns (loss tensor (1,), prediction (B,T_out,O))."""
        w, g = self.w, self.g
        B, T_in, _ = enc.shape
        T_out = dec_in.shape[1]
        bufs = self._buffers(B, T_in, T_out)
        ehs, ehT, ecT, eres = ops.lstm_seq_train(enc, w["enc_K"], w["enc_R"], w["enc_b"], act=self.act, impl=self.impl,
                                                 workspace=self.ws, out=bufs["enc"])
        dhs_, _, _, dres = ops.lstm_seq_train(dec_in, w["dec_K"], w["dec_R"], w["dec_b"], ehT, ecT, act=self.act,
                                              impl=self.impl, workspace=self.ws, out=bufs["dec"])
        # the DP weight rides on the loss gradient (everything downstream is linear in it); the loss lands in the
        # flat buffer's last slot
        if ops.dense_mse_head_supported(B * T_out, dhs_.shape[2], target.shape[2]):
            # few rows (the reference's batch: 320): Dense forward, loss, loss gradient, dX, dW and db as ONE launch
            y, d_hs, loss = ops.dense_mse_head(dhs_, w["dense_W"], w["dense_b"], target, "tanh", dW=g["dense_W"], db=g["dense_b"],
                                               loss=self.loss_slot, weight=grad_weight, scratch=self.scratch, dx=bufs["d_hs"], y=bufs["y"])
        else:
            y = ops.dense(dhs_, w["dense_W"], w["dense_b"], activation="tanh")
            # ... and the head's bias gradient (the column sums of dpre) comes from the same launch
            dpre, loss = ops.mse_dense_grad(y, target, "tanh", scratch=self.scratch, dpre=bufs["dpre"], loss=self.loss_slot,
                                            weight=grad_weight, db=g["dense_b"])
            d_hs, _, _ = ops.dense_bwd(dhs_, w["dense_W"], dpre, dW=g["dense_W"], need_db=False, scratch=self.scratch)
        if ops.lstm_seq_wgrad_pair_one_launch(B, T_in, T_out, dhs_.shape[2]):
            # few rows (the reference's batch): both recurrences back to back, then ALL six weight gradients as one launch
            bd = ops.lstm_seq_bwd(dec_in, w["dec_K"], w["dec_R"], dhs_, dres, h0=ehT, c0=ecT, dhs=d_hs, need_state_grads=True,
                                  act=self.act, dz=bufs["dz_dec"], scratch=self.bwd_scratch, need_weight_grads=False)
            ops.lstm_seq_bwd(enc, w["enc_K"], w["enc_R"], ehs, eres, dhT=bd["dh0"], dcT=bd["dc0"], act=self.act, dz=bufs["dz_enc"],
                             scratch=self.bwd_scratch, need_weight_grads=False)
            ops.lstm_seq_wgrad_pair((enc, ehs, None, bufs["dz_enc"], g["enc_K"], g["enc_R"], g["enc_b"]),
                                    (dec_in, dhs_, ehT, bufs["dz_dec"], g["dec_K"], g["dec_R"], g["dec_b"]), scratch=self.scratch)
            return loss, y
        bd = ops.lstm_seq_bwd(dec_in, w["dec_K"], w["dec_R"], dhs_, dres, h0=ehT, c0=ecT, dhs=d_hs, dK=g["dec_K"],
                              dR=g["dec_R"], db=g["dec_b"], need_state_grads=True, act=self.act, dz=bufs["dz_dec"],
                              scratch=self.bwd_scratch)
        self.grads_final("dec_K")     # decoder + head + loss: all-reduced under the encoder's BPTT
        ops.lstm_seq_bwd(enc, w["enc_K"], w["enc_R"], ehs, eres, dhT=bd["dh0"], dcT=bd["dc0"], dK=g["enc_K"],
                         dR=g["enc_R"], db=g["enc_b"], act=self.act, dz=bufs["dz_enc"], scratch=self.bwd_scratch)
        return loss, y

    def eval_loss(self, enc, dec_in, target):
        y = ops.seq2seq_teacher_forced(enc, dec_in, self.w, act=self.act, impl=self.impl, workspace=self.ws)
        _, loss = ops.mse_dense_grad(y, target, None, scratch=self.scratch)
        return loss


def self_fed_weight_order(add_residual_link=False, embed_frame_state_enc2dec=False, has_reconstruct_loss=False):
    """Parameter order of onelayer_tar_seq2seq (FoV_seq2seq_no_teac_forc.py:37-149) with its optional layers."""
    order = ("enc_K", "enc_R", "enc_b")
    if embed_frame_state_enc2dec:
        order += ("emb1_W", "emb1_b", "emb2_W", "emb2_b")
    order += ("dec_K", "dec_R", "dec_b")
    if has_reconstruct_loss:
        order += ("rec_K", "rec_R", "rec_b", "recd_W", "recd_b")
    order += ("dense_W", "dense_b")
    if add_residual_link:
        order += ("res_W", "res_b")
    return order


class SelfFedSeq2SeqTrainer(FlatParamTrainer):
    """Training step of the one-layer target-only model WITHOUT teacher forcing
    (mycode/FoV_seq2seq_no_teac_forc.py:37-149, `onelayer_tar_seq2seq`; Adam + MSE, :147): the decoder is unrolled
    T_out times on its own output.  Same layer kernels as the teacher-forced trainer, walked step by step: forward
    writes a time-major tape, backward joins the loss gradient of step t with the input gradient of step t+1, and
    every weight gradient is one product over all steps.

    decoder_no_init_state       the script's module flag (:29,98-99): step 0 starts from zero state (the encoder then
                                only matters through enc_last_out_as_dec_in / the reconstruction decoder)
    add_residual_link           cfg.add_residual_link (:70-72,103-107): y_t += residual_dense(decoder input)
    enc_last_out_as_dec_in      cfg.enc_last_out_as_dec_in (:75-78): decoder input = decoder_dense(encoder output)
    dense_activation            'tanh', or 'relu' under cfg.rescale_input (:60-63)
    embed_frame_state_enc2dec   cfg.embed_frame_state_enc2dec (:47-52): Dense(latent_dim, tanh) on the encoder's h and c
    has_reconstruct_loss        cfg.has_reconstruct_loss (:56-59,90-95,120-126,134-139): a second self-fed LSTM +
                                Dense(num_encoder_tokens, tanh) reconstructs input seconds; loss = MSE + MSE (weights 1, 1);
                                the target is then (B, T_out, O + F): prediction target | reconstruction target"""

    def __init__(self, weights, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda",
                 decoder_no_init_state=True, add_residual_link=False, enc_last_out_as_dec_in=False,
                 dense_activation="tanh", embed_frame_state_enc2dec=False, has_reconstruct_loss=False):
        self.act, self.impl = act, impl
        self.no_init, self.residual, self.enc_as_in = bool(decoder_no_init_state), bool(add_residual_link), bool(enc_last_out_as_dec_in)
        self.embed, self.recons = bool(embed_frame_state_enc2dec), bool(has_reconstruct_loss)
        self._alloc(weights, self_fed_weight_order(self.residual, self.embed, self.recons), optimizer, lr, device)
        self.dact = dense_activation

    def _dense(self, x, W, b, out=None, dact=None):
        dact = dact or self.dact
        y = ops.dense(x, W, b, activation="tanh" if dact == "tanh" else None, out=out)
        if dact == "relu":
            ops.act_fwd(y, "relu", out=y)
        return y

    # ---- one self-fed unrolled decoder: LSTM `lstm`_K/R/b + Dense `head`_W/b, y_t fed back as x_{t+1} ----
    def _unroll_forward(self, lstm, head, x0, h0, c0, T, dact, r=None):
        w, act, impl, ws = self.w, self.act, self.impl, self.ws
        B, O = x0.shape
        H = w[lstm + "_R"].shape[0]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        tp = {"Hs": e(T + 1, B, H), "Cs": e(T + 1, B, H),    # row t = state before step t
              "XA": e(T + 1, B, O),                           # row t = input x_t; row t+1 = y_t
              "A": e(T, B, O) if r is not None else None,     # Dense outputs before the residual is added
              "RES": e(T, B, 1, 5, H), "lstm": lstm, "head": head, "dact": dact, "r": r, "T": T}
        tp["XA"][0].copy_(x0)
        if h0 is None:
            tp["Hs"][0].zero_(); tp["Cs"][0].zero_()
        else:
            tp["Hs"][0].copy_(h0); tp["Cs"][0].copy_(c0)
        Hs, Cs, XA, A = tp["Hs"], tp["Cs"], tp["XA"], tp["A"]
        for t in range(T):
            ops.lstm_seq_train(XA[t].view(B, 1, O), w[lstm + "_K"], w[lstm + "_R"], w[lstm + "_b"], Hs[t], Cs[t], act=act, impl=impl,
                               workspace=ws, out=(Hs[t + 1].view(B, 1, H), None, Cs[t + 1], tp["RES"][t]))
            if r is not None:
                self._dense(Hs[t + 1], w[head + "_W"], w[head + "_b"], out=A[t], dact=dact)
                ops.act_bwd(r, r, base=A[t], activation=None, out=XA[t + 1])     # y_t = a_t + r
            else:
                self._dense(Hs[t + 1], w[head + "_W"], w[head + "_b"], out=XA[t + 1], dact=dact)
        return tp

    def _unroll_backward(self, tp, dloss_tm):
        """dloss_tm (T,B,O): dL/dy_t.  Accumulates the gradients of the unrolled LSTM and its head; -> (dx0, dh0, dc0, DY)."""
        w, g, act, sc, bsc = self.w, self.g, self.act, self.scratch, self.bwd_scratch
        lstm, head, dact, T = tp["lstm"], tp["head"], tp["dact"], tp["T"]
        Hs, Cs, XA = tp["Hs"], tp["Cs"], tp["XA"]
        _, B, O = XA.shape
        H = Hs.shape[2]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        Aout = tp["A"] if tp["r"] is not None else XA[1:]
        DY, DPRE, DZ = e(T, B, O), e(T, B, O), e(T, B, 4 * H)
        dh_rec = dc = dx_next = None
        for t in range(T - 1, -1, -1):
            if dx_next is None:
                DY[t].copy_(dloss_tm[t])
            else:                                                                  # x_{t+1} = y_t: feedback gradient joins
                ops.act_bwd(dx_next.reshape(B, O), Aout[t], base=dloss_tm[t], activation=None, out=DY[t])
            ops.act_bwd(DY[t], Aout[t], activation=dact, out=DPRE[t])
            dh_dense, _, _ = ops.dense_bwd(Hs[t + 1], w[head + "_W"], DPRE[t], need_dx=True, need_dW=False, need_db=False, scratch=sc)
            b = ops.lstm_seq_bwd(XA[t].view(B, 1, O), w[lstm + "_K"], w[lstm + "_R"], Hs[t + 1].view(B, 1, H), tp["RES"][t], h0=Hs[t],
                                 c0=Cs[t], dhs=dh_dense.reshape(B, 1, H), dhT=dh_rec, dcT=dc, need_dx=True, need_state_grads=True,
                                 act=act, dz=DZ[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh_rec, dc, dx_next = b["dh0"], b["dc0"], b["dx"]
        TB = T * B
        fl = lambda a, n: a.reshape(TB, n)
        ops.dense_bwd(fl(Hs[1:], H), w[head + "_W"], fl(DPRE, O), dW=g[head + "_W"], db=g[head + "_b"], need_dx=False, accumulate=True, scratch=sc)
        ops.dense_bwd(fl(XA[:T], O), w[lstm + "_K"], fl(DZ, 4 * H), dW=g[lstm + "_K"], db=g[lstm + "_b"], need_dx=False, accumulate=True, scratch=sc)
        ops.dense_bwd(fl(Hs[:T], H), w[lstm + "_R"], fl(DZ, 4 * H), dW=g[lstm + "_R"], need_db=False, need_dx=False, accumulate=True, scratch=sc)
        return dx_next.reshape(B, O), dh_rec, dc, DY

    @staticmethod
    def _sum(a, b):
        if a is None or b is None:
            return a if b is None else b
        return ops.act_bwd(b, b, base=a, activation=None)

    def forward_backward(self, enc, dec_in, target, grad_weight=1.0):
        """dec_in (B,1,O) is ignored under enc_last_out_as_dec_in.  -> (loss (1,), prediction (B,T_out,O)); with the
        reconstruction decoder target is (B,T_out,O+F) and the second return value (B,T_out,O+F) likewise."""
        w, g, act, impl, ws, sc, bsc = self.w, self.g, self.act, self.impl, self.ws, self.scratch, self.bwd_scratch
        B, T_in, F = enc.shape
        T_out = target.shape[1]
        O = w["dense_W"].shape[1]
        self.grad.zero_()
        # ---------------- forward ----------------
        ehs, ehT, ecT, eres = ops.lstm_seq_train(enc, w["enc_K"], w["enc_R"], w["enc_b"], act=act, impl=impl, workspace=ws)
        sh, sc_ = ehT, ecT
        if self.embed:
            sh = self._dense(ehT, w["emb1_W"], w["emb1_b"], dact="tanh")
            sc_ = self._dense(ecT, w["emb2_W"], w["emb2_b"], dact="tanh")
        x0 = self._dense(ehT, w["dense_W"], w["dense_b"]) if self.enc_as_in else dec_in.reshape(B, O)
        r = self._dense(x0, w["res_W"], w["res_b"]) if self.residual else None
        main = self._unroll_forward("dec", "dense", x0, None if self.no_init else sh, None if self.no_init else sc_, T_out, self.dact, r)
        out = main["XA"][1:].transpose(0, 1).contiguous()
        rec = None
        if self.recons:
            xr0 = self._dense(ehT, w["recd_W"], w["recd_b"], dact="tanh")
            rec = self._unroll_forward("rec", "recd", xr0, sh, sc_, T_out, "tanh")
            rec_out = rec["XA"][1:].transpose(0, 1).contiguous()
            tgt_main, tgt_rec = target[..., :O].contiguous(), target[..., O:].contiguous()
        else:
            tgt_main = target
        # ---------------- backward ----------------
        dloss, loss = ops.mse_dense_grad(out, tgt_main, None, scratch=sc)            # dL/dy, all steps
        dx0, dsh, dsc, DY = self._unroll_backward(main, dloss.transpose(0, 1).contiguous())
        if self.no_init:
            dsh = dsc = None
        d_ehT = None
        if self.residual:                                 # r is added to every step's output
            dr = ops.colsum(DY.reshape(T_out, B * O), scratch=sc).reshape(B, O)
            dpre_r = ops.act_bwd(dr, r, activation=self.dact)
            dxr, _, _ = ops.dense_bwd(x0, w["res_W"], dpre_r, dW=g["res_W"], db=g["res_b"], need_dx=True, accumulate=True, scratch=sc)
            dx0 = ops.act_bwd(dxr, dxr, base=dx0, activation=None)
        if self.enc_as_in:                                # x_0 = Dense(encoder output)
            dpre0 = ops.act_bwd(dx0, x0, activation=self.dact)
            d_ehT, _, _ = ops.dense_bwd(ehT, w["dense_W"], dpre0, dW=g["dense_W"], db=g["dense_b"], need_dx=True, accumulate=True, scratch=sc)
        if self.recons:
            dloss_r, loss_r = ops.mse_dense_grad(rec_out, tgt_rec, None, scratch=sc)
            loss = ops.act_bwd(loss_r, loss_r, base=loss, activation=None)          # loss weights [1, 1] (:137)
            dxr0, dsh_r, dsc_r, _ = self._unroll_backward(rec, dloss_r.transpose(0, 1).contiguous())
            dpre_x = ops.act_bwd(dxr0, rec["XA"][0], activation="tanh")
            d_x, _, _ = ops.dense_bwd(ehT, w["recd_W"], dpre_x, dW=g["recd_W"], db=g["recd_b"], need_dx=True, accumulate=True, scratch=sc)
            d_ehT = self._sum(d_ehT, d_x)
            dsh, dsc = self._sum(dsh, dsh_r), self._sum(dsc, dsc_r)
        d_ecT = None
        if dsh is not None:                               # the (embedded) encoder state reaches a decoder
            if self.embed:
                dp_h = ops.act_bwd(dsh, sh, activation="tanh")
                dp_c = ops.act_bwd(dsc, sc_, activation="tanh")
                dsh, _, _ = ops.dense_bwd(ehT, w["emb1_W"], dp_h, dW=g["emb1_W"], db=g["emb1_b"], need_dx=True, accumulate=True, scratch=sc)
                dsc, _, _ = ops.dense_bwd(ecT, w["emb2_W"], dp_c, dW=g["emb2_W"], db=g["emb2_b"], need_dx=True, accumulate=True, scratch=sc)
            d_ehT, d_ecT = self._sum(d_ehT, dsh), dsc
        if d_ehT is not None:                             # otherwise the encoder does not reach the loss (reference quirk)
            ops.lstm_seq_bwd(enc, w["enc_K"], w["enc_R"], ehs, eres, dhT=d_ehT, dcT=d_ecT, dK=g["enc_K"], dR=g["enc_R"],
                             db=g["enc_b"], act=act, accumulate=True, scratch=bsc)
        loss = self._weigh(loss, grad_weight)
        return loss, (torch.cat([out, rec_out], -1) if self.recons else out)



OTHERS_FUTURE_ORDER = ("enc_K", "enc_R", "enc_b", "oth_K", "oth_R", "oth_b", "flat_W", "flat_b", "dec_K", "dec_R", "dec_b",
                       "dense_W", "dense_b")


class OthersFutureConvLSTMTrainer(FlatParamTrainer):
    """Training step of the SECOND model of mycode/FoV_seq2seq_no_teac_forc.py (:420-486, fit at :553-558): the unrolled
    no-teacher-forcing decoder (seeded by the encoder's state) whose Dense(6, tanh) head sees concat[decoder output,
    Dense(latent_dim)(Flatten(ConvLSTM2D output t))], the ConvLSTM2D (filters = latent_dim, kernel (num_user-1, 3), 'same')
    running over the other users' future (B,T_out,num_user-1,fps,3) from zero state.  Adam + MSE.

    The branch does not depend on the decoder: its T_out steps run first (one fused cell launch per step, conv_kernels.hip),
    Flatten + Dense is ONE product over all steps, and its share of the head, s_t . dense_W[H:], is added per step inside the
    decoder's Dense launch (fov_dense_add_fwd).  Backward: the self-fed decoder in reverse (as SelfFedSeq2SeqTrainer), every
    weight gradient one product over all steps, then the branch's BPTT (gate backward + a forward convolution with the transposed
    recurrent kernel per step, weight gradients by conv2d_wgrad over the stacked steps), then the encoder's BPTT."""

    def __init__(self, weights, act="sigmoid", conv_act="hard_sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda"):
        self.act, self.conv_act, self.impl = act, conv_act, impl
        self._alloc(weights, OTHERS_FUTURE_ORDER, optimizer, lr, device)

    def branch_forward(self, others, tape=True):
        """others (B,T,U-1,fps,3) -> S (T*B, H) = Dense(Flatten(ConvLSTM2D output)) per step, time-major; + the tape."""
        w = self.w
        B, T, U1, Wd, C = others.shape
        H = w["oth_R"].shape[2]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        xs = others.permute(1, 0, 2, 3, 4).contiguous()
        hs, cs = e(T, B, U1, Wd, H), e(T, B, U1, Wd, H)
        gs = e(T, B, U1, Wd, 4 * H) if tape else None
        KR = torch.cat([w["oth_K"], w["oth_R"]], 2)            # [K ; R]: both convolutions of a step in one launch
        for t in range(T):
            if t > 0:
                ops.convlstm_cell(xs[t], hs[t - 1], KR, w["oth_b"], cs[t - 1], hs[t], self.conv_act, c_new=cs[t], gates=None if gs is None else gs[t])
            else:
                ops.convlstm_cell(xs[0], None, w["oth_K"], w["oth_b"], None, hs[0], self.conv_act, c_new=cs[0], gates=None if gs is None else gs[0])
        S = ops.dense(hs.reshape(T * B, U1 * Wd * H), w["flat_W"], w["flat_b"], activation=None)
        return S, {"xs": xs, "hs": hs, "cs": cs, "gs": gs}

    def decode(self, x0, h0, c0, S, T, tape=True):
        """The unrolled decoder: x0 (B,O), (h0, c0) the encoder's state, S (T*B,H).  -> tape with XA[1:] = y_t (T,B,O)."""
        w, act, impl, ws = self.w, self.act, self.impl, self.ws
        B, O = x0.shape
        H = w["dec_R"].shape[0]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        Wh, Wc = w["dense_W"][:H], w["dense_W"][H:]
        ADD = ops.matmul(S, Wc, scratch=self.scratch).reshape(T, B, O)      # the branch's share of the head's pre-activation
        tp = {"Hs": e(T + 1, B, H), "Cs": e(T + 1, B, H), "XA": e(T + 1, B, O), "RES": e(T, B, 1, 5, H) if tape else None, "T": T}
        tp["XA"][0].copy_(x0); tp["Hs"][0].copy_(h0); tp["Cs"][0].copy_(c0)
        Hs, Cs, XA = tp["Hs"], tp["Cs"], tp["XA"]
        for t in range(T):
            if tape:
                ops.lstm_seq_train(XA[t].view(B, 1, O), w["dec_K"], w["dec_R"], w["dec_b"], Hs[t], Cs[t], act=act, impl=impl, workspace=ws,
                                   out=(Hs[t + 1].view(B, 1, H), None, Cs[t + 1], tp["RES"][t]))
            else:
                _, hT, cT = ops.lstm_seq(XA[t].view(B, 1, O), w["dec_K"], w["dec_R"], w["dec_b"], Hs[t], Cs[t], act=act, impl=impl,
                                         return_sequences=False, workspace=ws)
                Hs[t + 1].copy_(hT); Cs[t + 1].copy_(cT)
            ops.dense_add(Hs[t + 1], Wh, w["dense_b"], ADD[t], activation="tanh", out=XA[t + 1])
        return tp

    def forward_backward(self, enc, others, dec_in, target, grad_weight=1.0):
        """enc (B,T_in,F), others (B,T_out,U-1,fps,3), dec_in (B,1,O), target (B,T_out,O) -> (loss (1,), prediction (B,T_out,O))."""
        w, g, act, impl, ws, sc, bsc = self.w, self.g, self.act, self.impl, self.ws, self.scratch, self.bwd_scratch
        B, T_in, F = enc.shape
        T = target.shape[1]
        O = w["dense_W"].shape[1]
        H = w["dec_R"].shape[0]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        self.grad.zero_()
        # ---------------- forward ----------------
        ehs, ehT, ecT, eres = ops.lstm_seq_train(enc, w["enc_K"], w["enc_R"], w["enc_b"], act=act, impl=impl, workspace=ws)
        S, bt = self.branch_forward(others)
        tp = self.decode(dec_in.reshape(B, O), ehT, ecT, S, T)
        Hs, Cs, XA = tp["Hs"], tp["Cs"], tp["XA"]
        out = XA[1:].transpose(0, 1).contiguous()
        # ---------------- backward: the self-fed decoder ----------------
        dloss, loss = ops.mse_dense_grad(out, target, None, scratch=sc)            # dL/dy, all steps
        dloss_tm = dloss.transpose(0, 1).contiguous()
        Wh, Wc = w["dense_W"][:H], w["dense_W"][H:]
        DY, DPRE, DZ = e(T, B, O), e(T, B, O), e(T, B, 4 * H)
        dh_rec = dc = dx_next = None
        for t in range(T - 1, -1, -1):
            if dx_next is None:
                DY[t].copy_(dloss_tm[t])
            else:                                                                  # x_{t+1} = y_t: the feedback gradient joins
                ops.act_bwd(dx_next.reshape(B, O), XA[t + 1], base=dloss_tm[t], activation=None, out=DY[t])
            ops.act_bwd(DY[t], XA[t + 1], activation="tanh", out=DPRE[t])
            dh_dense, _, _ = ops.dense_bwd(Hs[t + 1], Wh, DPRE[t], need_dx=True, need_dW=False, need_db=False, scratch=sc)
            b = ops.lstm_seq_bwd(XA[t].view(B, 1, O), w["dec_K"], w["dec_R"], Hs[t + 1].view(B, 1, H), tp["RES"][t], h0=Hs[t], c0=Cs[t],
                                 dhs=dh_dense.reshape(B, 1, H), dhT=dh_rec, dcT=dc, need_dx=True, need_state_grads=True, act=act,
                                 dz=DZ[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh_rec, dc, dx_next = b["dh0"], b["dc0"], b["dx"]
        TB = T * B
        fl = lambda a, n: a.reshape(TB, n)
        ops.dense_bwd(fl(Hs[1:], H), Wh, fl(DPRE, O), dW=g["dense_W"][:H], db=g["dense_b"], need_dx=False, scratch=sc)
        dS, _, _ = ops.dense_bwd(S, Wc, fl(DPRE, O), dW=g["dense_W"][H:], need_db=False, need_dx=True, scratch=sc)
        ops.dense_bwd(fl(XA[:T], O), w["dec_K"], fl(DZ, 4 * H), dW=g["dec_K"], db=g["dec_b"], need_dx=False, scratch=sc)
        ops.dense_bwd(fl(Hs[:T], H), w["dec_R"], fl(DZ, 4 * H), dW=g["dec_R"], need_db=False, need_dx=False, scratch=sc)
        # ---------------- the others' future branch ----------------
        xs, hs, cs, gs = bt["xs"], bt["hs"], bt["cs"], bt["gs"]
        U1, Wd = xs.shape[2], xs.shape[3]
        dHS, _, _ = ops.dense_bwd(hs.reshape(TB, U1 * Wd * H), w["flat_W"], dS, dW=g["flat_W"], db=g["flat_b"], need_dx=True, scratch=sc)
        dHS = dHS.reshape(T, B, U1, Wd, H)
        kh, kw = w["oth_K"].shape[:2]
        wtR = ops.conv2d_weight_transpose(w["oth_R"])
        edz = torch.empty_like(gs)
        dcl = torch.zeros((B, U1, Wd, H), dtype=torch.float32, device=self.device)
        dhr = None
        for t in range(T - 1, -1, -1):
            dh = dHS[t] if dhr is None else ops.act_bwd(dhr, dhr, base=dHS[t], activation=None)
            dz = ops.convlstm_gates_bwd(dh, dcl, gs[t], cs[t - 1] if t > 0 else None, cs[t], self.conv_act, dz=edz[t])
            if t > 0:
                dhr = ops.conv2d(dz, wtR)
        ops.conv2d_wgrad(xs, edz, kh, kw, dw=g["oth_K"], scratch=sc)
        if T > 1:
            ops.conv2d_wgrad(hs[:T - 1], edz[1:], kh, kw, dw=g["oth_R"], scratch=sc)
        ops.colsum(edz, out=g["oth_b"], scratch=sc)
        # ---------------- the encoder (its state seeds the decoder) ----------------
        ops.lstm_seq_bwd(enc, w["enc_K"], w["enc_R"], ehs, eres, dhT=dh_rec, dcT=dc, dK=g["enc_K"], dR=g["enc_R"], db=g["enc_b"], act=act,
                         scratch=bsc)
        loss = self._weigh(loss, grad_weight)
        return loss, out


def stacked_weight_order(num_layers):
    return tuple("%s%d_%s" % (side, l, n) for side in ("enc", "dec") for l in range(num_layers) for n in ("K", "R", "b")) + \
        ("dense_W", "dense_b")


class StackedSeq2SeqTrainer(FlatParamTrainer):
    """Teacher-forced training step of the L-layer target-only seq2seq (mycode/Fov_seq2seq_2layers.py:232-272,332-343 and
    3layers.py:222-300): encoder layer l hands its final (h, c) to decoder layer l, every layer returns its sequence to
    the next, Dense(6, tanh) on the top decoder layer; Adam + MSE.  Same layer kernels as the one-layer trainer, one
    forward-with-reserve and one BPTT launch per layer."""

    defer_reduces = True   # gradients are written by the library's weight-gradient entry points only


    def __init__(self, weights, num_layers, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda"):
        self.L, self.act, self.impl = int(num_layers), act, impl
        self._alloc(weights, stacked_weight_order(self.L), optimizer, lr, device)

    def forward_backward(self, enc, dec_in, target, grad_weight=1.0):
        w, g, act, impl, ws, L = self.w, self.g, self.act, self.impl, self.ws, self.L
        etape, dtape = [], []
        inp = enc
        for l in range(L):
            hs, hT, cT, res = ops.lstm_seq_train(inp, w["enc%d_K" % l], w["enc%d_R" % l], w["enc%d_b" % l], act=act, impl=impl, workspace=ws)
            etape.append((inp, hs, res, hT, cT))
            inp = hs
        inp = dec_in
        for l in range(L):
            hs, _, _, res = ops.lstm_seq_train(inp, w["dec%d_K" % l], w["dec%d_R" % l], w["dec%d_b" % l], etape[l][3], etape[l][4],
                                               act=act, impl=impl, workspace=ws)
            dtape.append((inp, hs, res))
            inp = hs
        y = ops.dense(inp, w["dense_W"], w["dense_b"], activation="tanh")
        dpre, loss = ops.mse_dense_grad(y, target, "tanh", scratch=self.scratch)
        d_cur, _, _ = ops.dense_bwd(inp, w["dense_W"], dpre, dW=g["dense_W"], db=g["dense_b"], scratch=self.scratch)
        dstate = [None] * L
        for l in range(L - 1, -1, -1):
            x_l, hs, res = dtape[l]
            bd = ops.lstm_seq_bwd(x_l, w["dec%d_K" % l], w["dec%d_R" % l], hs, res, h0=etape[l][3], c0=etape[l][4], dhs=d_cur,
                                  dK=g["dec%d_K" % l], dR=g["dec%d_R" % l], db=g["dec%d_b" % l], need_dx=(l > 0),
                                  need_state_grads=True, act=act, scratch=self.bwd_scratch)
            d_cur, dstate[l] = bd["dx"], (bd["dh0"], bd["dc0"])
        d_cur = None
        for l in range(L - 1, -1, -1):
            x_l, hs, res, _, _ = etape[l]
            be = ops.lstm_seq_bwd(x_l, w["enc%d_K" % l], w["enc%d_R" % l], hs, res, dhs=d_cur, dhT=dstate[l][0], dcT=dstate[l][1],
                                  dK=g["enc%d_K" % l], dR=g["enc%d_R" % l], db=g["enc%d_b" % l], need_dx=(l > 0), act=act,
                                  scratch=self.bwd_scratch)
            d_cur = be["dx"]
        loss = self._weigh(loss, grad_weight)
        return loss, y



def others_context_order(mode):
    base = tuple("%s_%s" % (n, p) for n in ("enc1", "enc2", "dec1", "dec2") for p in ("K", "R", "b")) + ("dense_W", "dense_b")
    if mode == "others_mlp":
        return base + ("oth_W1", "oth_b1", "oth_W2", "oth_b2")
    if mode == "others_lstm":
        return base + tuple("ol%d%s_%s" % (j, d, p) for j in (1, 2) for d in ("f", "b") for p in ("K", "R", "b"))
    return base


class OthersContextTrainer(FlatParamTrainer):
    """Training step of the other decoder heads of mycode/given_others_gt_mean_var_seq2seq.py (2+2-layer model, no teacher
    forcing, Adam + MSE): `target_user_only` (:219-220), `others_mlp` (:153-156,223-233), `others_lstm` (two Bidirectional
    LSTMs over the others' future mu/var, :157-166,234-240).  The others only enter through a per-step context that
    does not depend on the decoder, so its projection through decoder_dense is ONE product hoisted out of the unrolled
    loop (y_t = tanh(h2_t W_h + [ctx_t W_c + b])) and so is its gradient; the decoder is walked step by step on the
    layer kernels, weight gradients are one product over all steps."""

    def __init__(self, weights, mode, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda"):
        assert mode in ("target_user_only", "others_mlp", "others_lstm")
        self.mode, self.act, self.impl = mode, act, impl
        self._alloc(weights, others_context_order(mode), optimizer, lr, device)

    # ---- context: forward keeps what its backward needs ----
    def _bi_fwd(self, x, j, init):
        w, out = self.w, {}
        for d, xin in (("f", x), ("b", torch.flip(x, (1,)))):
            n = "ol%d%s" % (j, d)
            h0, c0 = (None, None) if init is None else init[d]
            hs, hT, cT, res = ops.lstm_seq_train(xin, w[n + "_K"], w[n + "_R"], w[n + "_b"], h0, c0, act=self.act, impl=self.impl,
                                                 workspace=self.ws)
            out[d] = (xin, hs, res, hT, cT, h0, c0)
        seq = torch.cat([out["f"][1], torch.flip(out["b"][1], (1,))], 2)
        return seq, out

    def _bi_bwd(self, dseq, tape, j, dstate, need_dx):
        """dseq (B,T,2H) -> gradient w.r.t. the layer input (or None) and w.r.t. its initial states {dir: (dh0, dc0)}."""
        w, g = self.w, self.g
        H = dseq.shape[2] // 2
        dx, dinit = None, {}
        for d in ("f", "b"):
            n = "ol%d%s" % (j, d)
            xin, hs, res, _, _, h0, c0 = tape[d]
            dhs = dseq[..., :H].contiguous() if d == "f" else torch.flip(dseq[..., H:], (1,)).contiguous()
            dhT, dcT = (None, None) if dstate is None else dstate[d]
            b = ops.lstm_seq_bwd(xin, w[n + "_K"], w[n + "_R"], hs, res, h0=h0, c0=c0, dhs=dhs, dhT=dhT, dcT=dcT, dK=g[n + "_K"],
                                 dR=g[n + "_R"], db=g[n + "_b"], need_dx=need_dx, need_state_grads=(h0 is not None), act=self.act,
                                 accumulate=True, scratch=self.bwd_scratch)
            dinit[d] = (b["dh0"], b["dc0"])
            if need_dx:
                dxd = b["dx"] if d == "f" else torch.flip(b["dx"], (1,)).contiguous()
                dx = dxd if dx is None else ops.act_bwd(dxd, dxd, base=dx, activation=None)
        return dx, dinit

    def forward_backward(self, enc, others, dec0, target, grad_weight=1.0):
        """others (B,T_out,U-1,6) (ignored by 'target_user_only').  -> (loss (1,), prediction (B,T_out,O))."""
        w, g, act, impl, ws, sc, bsc = self.w, self.g, self.act, self.impl, self.ws, self.scratch, self.bwd_scratch
        B, T_in, _ = enc.shape
        T_out, O = target.shape[1], target.shape[2]
        H = w["enc1_R"].shape[0]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        self.grad.zero_()
        Cc = w["dense_W"].shape[0] - H
        W_c, W_h = w["dense_W"][:Cc], w["dense_W"][Cc:].contiguous()
        gW_c, gW_h = g["dense_W"][:Cc], g["dense_W"][Cc:]
        # ---------------- context ----------------
        ctx = ctape = None
        if self.mode == "others_mlp":
            o2 = others.reshape(B * T_out, -1)
            a1 = ops.dense(o2, w["oth_W1"], w["oth_b1"], activation=None)
            ops.act_fwd(a1, "relu", out=a1)
            ctx = ops.dense(a1, w["oth_W2"], w["oth_b2"], activation=None)
            ops.act_fwd(ctx, "relu", out=ctx)
            ctape = (o2, a1)
        elif self.mode == "others_lstm":
            o3 = others.reshape(B, T_out, -1)
            s1, t1 = self._bi_fwd(o3, 1, None)
            s2, t2 = self._bi_fwd(s1, 2, {d: (t1[d][3], t1[d][4]) for d in ("f", "b")})
            ctx, ctape = s2.reshape(B * T_out, 2 * H), (t1, t2)
        if ctx is not None:      # [ctx_t W_c + b] for every step at once, rows (b, t)
            ctx_proj = ops.dense(ctx, W_c.contiguous(), w["dense_b"], activation=None).reshape(B, T_out, O)
        # ---------------- encoder + unrolled decoder ----------------
        H1, C1, H2, C2 = e(T_out + 1, B, H), e(T_out + 1, B, H), e(T_out + 1, B, H), e(T_out + 1, B, H)
        hs1, _, _, res1 = ops.lstm_seq_train(enc, w["enc1_K"], w["enc1_R"], w["enc1_b"], act=act, impl=impl, workspace=ws,
                                             out=(e(B, T_in, H), H1[0], C1[0], e(B, T_in, 5, H)))
        hs2, _, _, res2 = ops.lstm_seq_train(hs1, w["enc2_K"], w["enc2_R"], w["enc2_b"], act=act, impl=impl, workspace=ws,
                                             out=(e(B, T_in, H), H2[0], C2[0], e(B, T_in, 5, H)))
        XY = e(T_out + 1, B, O)
        R1, R2 = e(T_out, B, 1, 5, H), e(T_out, B, 1, 5, H)
        XY[0].copy_(dec0.reshape(B, O))
        for t in range(T_out):
            ops.lstm_seq_train(XY[t].view(B, 1, O), w["dec1_K"], w["dec1_R"], w["dec1_b"], H1[t], C1[t], act=act, impl=impl,
                               workspace=ws, out=(H1[t + 1].view(B, 1, H), None, C1[t + 1], R1[t]))
            ops.lstm_seq_train(H1[t + 1].view(B, 1, H), w["dec2_K"], w["dec2_R"], w["dec2_b"], H2[t], C2[t], act=act, impl=impl,
                               workspace=ws, out=(H2[t + 1].view(B, 1, H), None, C2[t + 1], R2[t]))
            if ctx is None:
                ops.dense(H2[t + 1], W_h, w["dense_b"], activation="tanh", out=XY[t + 1])
            else:
                ops.dense_add(H2[t + 1], W_h, None, ctx_proj[:, t], activation="tanh", out=XY[t + 1])
        out = XY[1:].transpose(0, 1).contiguous()
        # ---------------- backward ----------------
        dloss, loss = ops.mse_dense_grad(out, target, None, scratch=sc)
        dloss_tm = dloss.transpose(0, 1).contiguous()
        DY, DPRE, DZ1, DZ2 = e(T_out, B, O), e(T_out, B, O), e(T_out, B, 4 * H), e(T_out, B, 4 * H)
        dh1 = dc1 = dh2 = dc2 = dx_next = None
        for t in range(T_out - 1, -1, -1):
            if dx_next is None:
                DY[t].copy_(dloss_tm[t])
            else:
                ops.act_bwd(dx_next.reshape(B, O), XY[t + 1], base=dloss_tm[t], activation=None, out=DY[t])
            ops.act_bwd(DY[t], XY[t + 1], activation="tanh", out=DPRE[t])
            dh2_dense, _, _ = ops.dense_bwd(H2[t + 1], W_h, DPRE[t], need_dx=True, need_dW=False, need_db=False, scratch=sc)
            b2 = ops.lstm_seq_bwd(H1[t + 1].view(B, 1, H), w["dec2_K"], w["dec2_R"], H2[t + 1].view(B, 1, H), R2[t], h0=H2[t],
                                  c0=C2[t], dhs=dh2_dense.reshape(B, 1, H), dhT=dh2, dcT=dc2, need_dx=True, need_state_grads=True,
                                  act=act, dz=DZ2[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh2, dc2 = b2["dh0"], b2["dc0"]
            b1 = ops.lstm_seq_bwd(XY[t].view(B, 1, O), w["dec1_K"], w["dec1_R"], H1[t + 1].view(B, 1, H), R1[t], h0=H1[t],
                                  c0=C1[t], dhs=b2["dx"], dhT=dh1, dcT=dc1, need_dx=(t > 0), need_state_grads=True, act=act,
                                  dz=DZ1[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh1, dc1, dx_next = b1["dh0"], b1["dc0"], b1["dx"]
        TB = T_out * B
        fl = lambda a, n: a.reshape(TB, n)
        ops.dense_bwd(fl(H2[1:], H), W_h, fl(DPRE, O), dW=gW_h, db=g["dense_b"], need_dx=False, accumulate=True, scratch=sc)
        ops.dense_bwd(fl(H1[1:], H), w["dec2_K"], fl(DZ2, 4 * H), dW=g["dec2_K"], db=g["dec2_b"], need_dx=False, accumulate=True, scratch=sc)
        ops.dense_bwd(fl(H2[:T_out], H), w["dec2_R"], fl(DZ2, 4 * H), dW=g["dec2_R"], need_db=False, need_dx=False, accumulate=True, scratch=sc)
        ops.dense_bwd(fl(XY[:T_out], O), w["dec1_K"], fl(DZ1, 4 * H), dW=g["dec1_K"], db=g["dec1_b"], need_dx=False, accumulate=True, scratch=sc)
        ops.dense_bwd(fl(H1[:T_out], H), w["dec1_R"], fl(DZ1, 4 * H), dW=g["dec1_R"], need_db=False, need_dx=False, accumulate=True, scratch=sc)
        if ctx is not None:      # the context's half of decoder_dense and the context module, rows (b, t)
            dpre_bt = DPRE.transpose(0, 1).contiguous().reshape(B * T_out, O)
            dctx, _, _ = ops.dense_bwd(ctx, W_c.contiguous(), dpre_bt, dW=gW_c, need_db=False, need_dx=True, accumulate=True, scratch=sc)
            if self.mode == "others_mlp":
                o2, a1 = ctape
                d2 = ops.act_bwd(dctx, ctx, activation="relu")
                da1, _, _ = ops.dense_bwd(a1, w["oth_W2"], d2, dW=g["oth_W2"], db=g["oth_b2"], accumulate=True, scratch=sc)
                d1 = ops.act_bwd(da1, a1, activation="relu")
                ops.dense_bwd(o2, w["oth_W1"], d1, dW=g["oth_W1"], db=g["oth_b1"], need_dx=False, accumulate=True, scratch=sc)
            else:
                t1, t2 = ctape
                ds1, dinit = self._bi_bwd(dctx.reshape(B, T_out, 2 * H), t2, 2, None, need_dx=True)
                self._bi_bwd(ds1, t1, 1, dinit, need_dx=False)
        e2 = ops.lstm_seq_bwd(hs1, w["enc2_K"], w["enc2_R"], hs2, res2, dhT=dh2, dcT=dc2, dK=g["enc2_K"], dR=g["enc2_R"],
                              db=g["enc2_b"], need_dx=True, act=act, accumulate=True, scratch=bsc)
        ops.lstm_seq_bwd(enc, w["enc1_K"], w["enc1_R"], hs1, res1, dhs=e2["dx"], dhT=dh1, dcT=dc1, dK=g["enc1_K"], dR=g["enc1_R"],
                         db=g["enc1_b"], act=act, accumulate=True, scratch=bsc)
        loss = self._weigh(loss, grad_weight)
        return loss, out



_SINGLE_ORDER = ("K", "R", "b", "dense_W", "dense_b")


class SingleLSTMTrainer(FlatParamTrainer):
    """Training step of the single-layer Keras model of mycode/lstm_keras.py: ONE LSTM from zero state + Dense(6, tanh)
    per step, Adam + MSE (:59-83,214-241).  Two unrollings:
      per-step   (1st part, :70-80): the T input seconds are consumed one per step - LSTM(return_sequences) + Dense;
      unrolled   (2nd part, :218-240): ONE input second, `predict_step` steps.  Without re-feed the same input is shown
                 to every step; under cfg.predict_mean_var and cfg.sample_and_refeed the next input is a second SAMPLED
                 around the prediction: K.random_normal(mean = mu, stddev = var) per axis (the variance is handed to
                 `stddev`, :39-44), the three axes concatenated planar [x*fps | y*fps | z*fps] (:147-149); the gradient
                 flows through the sample (reparameterisation).  `noise` (T_out-1, B, 3*fps) is given by the caller."""

    def __init__(self, weights, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda", unrolled=False,
                 sample_and_refeed=False):
        self.act, self.impl = act, impl
        self._alloc(weights, _SINGLE_ORDER, optimizer, lr, device)
        self.unrolled, self.refeed = bool(unrolled), bool(unrolled and sample_and_refeed)

    def _sequence(self, x_seq, target):
        w, g = self.w, self.g
        hs, _, _, res = ops.lstm_seq_train(x_seq, w["K"], w["R"], w["b"], act=self.act, impl=self.impl, workspace=self.ws)
        y = ops.dense(hs, w["dense_W"], w["dense_b"], activation="tanh")
        dpre, loss = ops.mse_dense_grad(y, target, "tanh", scratch=self.scratch)
        d_hs, _, _ = ops.dense_bwd(hs, w["dense_W"], dpre, dW=g["dense_W"], db=g["dense_b"], scratch=self.scratch)
        ops.lstm_seq_bwd(x_seq, w["K"], w["R"], hs, res, dhs=d_hs, dK=g["K"], dR=g["R"], db=g["b"], act=self.act,
                         scratch=self.bwd_scratch)
        return loss, y

    def forward_backward(self, x, target, noise=None, grad_weight=1.0):
        """per-step: x (B,T,F), target (B,T,O).  unrolled: x (B,1,F), target (B,T_out,O); with the sampled re-feed
        (F = 3*fps, O = 6) `noise` (T_out-1, B, F) standard normal, drawn here when not given.  -> (loss (1,), prediction)."""
        w, g, act, impl, ws, sc, bsc = self.w, self.g, self.act, self.impl, self.ws, self.scratch, self.bwd_scratch
        B, T_out, O = target.shape
        if not self.unrolled:
            loss, y = self._sequence(x, target)
        elif not self.refeed:
            loss, y = self._sequence(x.expand(B, T_out, x.shape[2]).contiguous(), target)
        else:
            if noise is None:
                noise = torch.randn((T_out - 1, B, x.shape[2]), dtype=torch.float32, device=self.device)
            F, H = x.shape[2], w["R"].shape[0]
            assert O == 6 and F % 3 == 0 and tuple(noise.shape) == (T_out - 1, B, F)
            e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
            self.grad.zero_()
            XS, Hs, Cs, Y, RES = e(T_out, B, F), e(T_out + 1, B, H), e(T_out + 1, B, H), e(T_out, B, O), e(T_out, B, 1, 5, H)
            XS[0].copy_(x.reshape(B, F))
            Hs[0].zero_(); Cs[0].zero_()
            stats = []
            for t in range(T_out):
                ops.lstm_seq_train(XS[t].view(B, 1, F), w["K"], w["R"], w["b"], Hs[t], Cs[t], act=act, impl=impl, workspace=ws,
                                   out=(Hs[t + 1].view(B, 1, H), None, Cs[t + 1], RES[t]))
                ops.dense(Hs[t + 1], w["dense_W"], w["dense_b"], activation="tanh", out=Y[t])
                if t < T_out - 1:
                    stats.append((Y[t, :, :3].contiguous(), Y[t, :, 3:].contiguous()))
                    ops.sample_refeed(stats[t][0], stats[t][1], noise[t], out=XS[t + 1], std="var", planar=True)
            y = Y.transpose(0, 1).contiguous()
            dloss, loss = ops.mse_dense_grad(y, target, None, scratch=sc)
            dloss_tm = dloss.transpose(0, 1).contiguous()
            DY, DPRE, DZ = e(T_out, B, O), e(T_out, B, O), e(T_out, B, 4 * H)
            dh_rec = dc = dstat = None
            for t in range(T_out - 1, -1, -1):
                if dstat is None:
                    DY[t].copy_(dloss_tm[t])
                else:
                    ops.act_bwd(dstat, dstat, base=dloss_tm[t], activation=None, out=DY[t])
                ops.act_bwd(DY[t], Y[t], activation="tanh", out=DPRE[t])
                dh_dense, _, _ = ops.dense_bwd(Hs[t + 1], w["dense_W"], DPRE[t], need_dx=True, need_dW=False, need_db=False, scratch=sc)
                b = ops.lstm_seq_bwd(XS[t].view(B, 1, F), w["K"], w["R"], Hs[t + 1].view(B, 1, H), RES[t], h0=Hs[t], c0=Cs[t],
                                     dhs=dh_dense.reshape(B, 1, H), dhT=dh_rec, dcT=dc, need_dx=(t > 0), need_state_grads=True,
                                     act=act, dz=DZ[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
                dh_rec, dc = b["dh0"], b["dc0"]
                if t > 0:
                    dmu, dvar = e(B, 3), e(B, 3)
                    ops.sample_refeed_bwd(b["dx"].view(B, F), stats[t - 1][1], noise[t - 1], dmu, dvar, std="var", planar=True,
                                          accumulate=False)
                    dstat = torch.cat([dmu, dvar], 1)
            TB = T_out * B
            ops.dense_bwd(Hs[1:].reshape(TB, H), w["dense_W"], DPRE.reshape(TB, O), dW=g["dense_W"], db=g["dense_b"], need_dx=False,
                          accumulate=True, scratch=sc)
            ops.dense_bwd(XS.reshape(TB, F), w["K"], DZ.reshape(TB, 4 * H), dW=g["K"], db=g["b"], need_dx=False, accumulate=True, scratch=sc)
            ops.dense_bwd(Hs[:T_out].reshape(TB, H), w["R"], DZ.reshape(TB, 4 * H), dW=g["R"], need_db=False, need_dx=False,
                          accumulate=True, scratch=sc)
        loss = self._weigh(loss, grad_weight)
        return loss, y



_MIX_ORDER = ("enc1_K", "enc1_R", "enc1_b", "enc2_K", "enc2_R", "enc2_b", "dec1_K", "dec1_R", "dec1_b",
              "dec2_K", "dec2_R", "dec2_b", "dense_W", "dense_b", "mix_W", "mix_b")


class OthersMixingTrainer(FlatParamTrainer):
    """Training step of the 2+2-layer others-mixing model WITHOUT teacher forcing
    (mycode/given_others_gt_mean_var_seq2seq.py:203-308: Adam + MSE on the unrolled decoder whose output
    is fed back).  Forward of the unrolled decoder: ONE persistent launch (fov_mix_decoder_fwd, H = 256) that
    writes the time-major tape, or step-wise layer calls for other shapes.  Backward: the decoder is walked
    step by step (data path only: mixing head, layer 2, layer 1; the feedback path's gradient is the dx of the
    first decoder layer), then every weight gradient is one product over all steps.  Gradients accumulate into
    ONE flat buffer (one all-reduce under DP)."""

    defer_reduces = True   # gradients are written by the library's weight-gradient entry points only


    fused_decoder = True       # H = 256: forward of the unrolled decoder as ONE launch; False = step-wise calls
    fused_decoder_bwd = True   # H = 256: BPTT through the unrolled decoder as ONE launch; False = step-wise calls

    def __init__(self, weights, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda", dtype="f32"):
        """dtype 'bf16' (BASELINE configs[4], H = 256): forward and backward matrix products take bf16 operands on the
        matrix cores with fp32 accumulation; gates, cell state, tapes, gradients' accumulation, the flat parameter
        buffer and the optimizer stay fp32 (fp32 master weights)."""
        assert dtype in ("f32", "bf16")
        self.act, self.impl, self.dtype = act, impl, dtype
        self._alloc(weights, _MIX_ORDER, optimizer, lr, device)
        if dtype == "bf16":
            assert self.w["enc1_R"].shape[0] == 256, "the bf16 path is built for H = 256"
        self.ws_bwd = ops.Workspace()   # granule mailboxes of the fused decoder backward

    def forward_backward(self, enc, others, dec0, target, grad_weight=1.0):
        w, g, act, impl, ws, sc, bsc = self.w, self.g, self.act, self.impl, self.ws, self.scratch, self.bwd_scratch
        B, T_in, _ = enc.shape
        T_out = others.shape[1]
        H = w["enc1_R"].shape[0]
        O = w["dense_W"].shape[1]
        n_oth = w["mix_W"].shape[0] - O
        Wm_o, Wm_p = w["mix_W"][:n_oth], w["mix_W"][n_oth:]
        acc = False               # every gradient below is written exactly once: nothing to zero, nothing to accumulate
        # ---------------- forward (keeping what the backward needs) ----------------
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        # decoder state tapes, time-major: row 0 = the encoder's final state (written there by the encoder kernels),
        # row t+1 = state after decoder step t
        H1, C1 = e(T_out + 1, B, H), e(T_out + 1, B, H)
        H2, C2 = e(T_out + 1, B, H), e(T_out + 1, B, H)
        dt = self.dtype
        # What depends on the inputs and the weights alone - the others' projection, the decoder's first input row, the fp32
        # fused kernels' packed copy of dec2_K - runs on the SIDE stream while the encoder layers run here (four launches of
        # 5-7 us each that used to sit between the encoder and the decoder).  Outputs are allocated on this stream.
        fused = self.fused_decoder and ops.mix_decoder_supported(H, O)
        fused_bwd = self.fused_decoder_bwd and ops.mix_decoder_supported(H, O)
        XM = e(T_out + 1, B, O)                               # row t = decoder input x_t, row t+1 = output m_t
        oth_proj_buf = e(B * T_out, O)
        oth_flat = others.reshape(B * T_out, n_oth)
        Wm_o_c, Wm_p_c = Wm_o.contiguous(), Wm_p.contiguous()

        def input_side_work():
            ops.dense(oth_flat, Wm_o_c, w["mix_b"], activation=None, out=oth_proj_buf)
            XM[0].copy_(dec0.reshape(B, O))
            if dt != "bf16" and (fused or fused_bwd):
                ops.mix_decoder_prepack(w["dec2_K"], B, H, ws if fused else None, self.ws_bwd if fused_bwd else None)
        side_in = self._wgrad_side_stream() if (B, T_in, T_out, dt) in self._side_warm else None
        if side_in is not None:
            side_in.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_in):
                input_side_work()
        stacked = dt == "bf16" and impl != "generic" and ops.lstm_stack2_bf16_supported(B, T_in, enc.shape[2], H)
        if stacked:   # both encoder layers as ONE wavefront launch (layer 2 one step behind layer 1 on the same CUs): same tensors
            (hs1, h1, c1, res1), (hs2, h2, c2, res2) = ops.lstm_stack2_bf16(
                enc, (w["enc1_K"], w["enc1_R"], w["enc1_b"]), (w["enc2_K"], w["enc2_R"], w["enc2_b"]), act=act, workspace=ws,
                out1=(e(B, T_in, H), H1[0], C1[0], e(B, T_in, 5, H)), out2=(e(B, T_in, H), H2[0], C2[0], e(B, T_in, 5, H)))
        else:
            hs1, h1, c1, res1 = ops.lstm_seq_train(enc, w["enc1_K"], w["enc1_R"], w["enc1_b"], act=act, impl=impl, workspace=ws,
                                                   out=(e(B, T_in, H), H1[0], C1[0], e(B, T_in, 5, H)), dtype=dt)
        if stacked:
            pass
        elif H == 256 and impl != "generic":   # layer 2 over the 256-wide sequence: K2 and R2 register-resident
            hs2, h2, c2, res2 = ops.lstm_seq_train(hs1, w["enc2_K"], w["enc2_R"], w["enc2_b"], act=act, impl=impl, workspace=ws,
                                                   out=(e(B, T_in, H), H2[0], C2[0], e(B, T_in, 5, H)), dtype=dt)
        else:
            zx2 = ops.matmul(hs1.reshape(B * T_in, H), w["enc2_K"], scratch=sc).reshape(B, T_in, 4 * H)
            res2 = torch.empty((B, T_in, 5, H), dtype=torch.float32, device=self.device)
            hs2, h2, c2 = ops.lstm_seq_zx(zx2, w["enc2_R"], w["enc2_b"], act=act, impl=impl, workspace=ws, reserve=res2,
                                          out=(e(B, T_in, H), H2[0], C2[0]))
        if side_in is None:
            input_side_work()
        else:
            torch.cuda.current_stream().wait_stream(side_in)
        oth_proj = oth_proj_buf.reshape(B, T_out, O)
        # Decoder tape, time-major: step t reads row t of the "previous state" stacks and writes row t+1, so the
        # stacked rows are exactly the operands of the per-layer weight-gradient products formed after the loop.
        X, M = XM[:T_out], XM[1:]
        R1, R2 = e(T_out, B, 1, 5, H), e(T_out, B, 1, 5, H)   # reserves (i,f,g,o,c) of every step
        P = e(T_out, B, O)
        if fused:   # the whole unrolled forward in one persistent launch, writing the same tape
            ops.mix_decoder(X[0], H1[0], C1[0], H2[0], C2[0], oth_proj, w, Wm_p_c, T_out, act=act, workspace=ws, out=M,
                            train={"P": P, "H1": H1[1:], "C1": C1[1:], "H2": H2[1:], "C2": C2[1:],
                                   "res1": R1.view(T_out, B, 5, H), "res2": R2.view(T_out, B, 5, H)}, dtype=dt)
        for t in range(0 if fused else T_out):
            # every kernel writes straight into its row of the tape: no copies inside the loop
            ops.lstm_seq_train(X[t].view(B, 1, O), w["dec1_K"], w["dec1_R"], w["dec1_b"], H1[t], C1[t], act=act, impl=impl,
                               workspace=ws, out=(H1[t + 1].view(B, 1, H), None, C1[t + 1], R1[t]))
            zx = ops.matmul(H1[t + 1], w["dec2_K"], scratch=sc).reshape(B, 1, 4 * H)
            ops.lstm_seq_zx(zx, w["dec2_R"], w["dec2_b"], H2[t], C2[t], act=act, impl=impl, workspace=ws, reserve=R2[t],
                            out=(H2[t + 1].view(B, 1, H), None, C2[t + 1]))
            ops.mix_head_fwd(H2[t + 1], w["dense_W"], w["dense_b"], Wm_p_c, oth_proj[:, t], P[t], M[t])
        out = M.transpose(0, 1)                                                   # (B,T_out,O) view of the time-major tape
        # ---------------- backward ----------------
        # dL/d(pre-tanh of the mixing layer) from the loss, for every step at once, straight in the tape's time-major
        # order; the DP weight rides on it (everything downstream is linear in it), the loss lands in the flat
        # buffer's last slot
        dloss_tm, loss = ops.mse_dense_grad(M, target, "tanh", scratch=sc, loss=self.loss_slot, weight=grad_weight,
                                            time_major=True)                       # (T_out,B,O), per-step rows
        dpre_all, dpre_p_all = e(T_out, B, O), e(T_out, B, O)                      # written by the head backward
        DZ1, DZ2 = e(T_out, B, 4 * H), e(T_out, B, 4 * H)
        dh1_rec = dc1 = dh2_rec = dc2 = None
        dx_next = None
        if fused_bwd:   # BPTT through the whole unrolled decoder (head, layer 2, layer 1, feedback) in one launch
            dh1_rec, dc1, dh2_rec, dc2 = e(B, H), e(B, H), e(B, H), e(B, H)
            ops.mix_decoder_bwd(M, P, dloss_tm, R1.view(T_out, B, 5, H), R2.view(T_out, B, 5, H), C1, C2, w, Wm_p_c,
                                {"DZ1": DZ1, "DZ2": DZ2, "dpre_m": dpre_all, "dpre_p": dpre_p_all, "dh1_0": dh1_rec,
                                 "dc1_0": dc1, "dh2_0": dh2_rec, "dc2_0": dc2}, act=act, workspace=self.ws_bwd, dtype=dt)
        for t in range(-1 if fused_bwd else T_out - 1, -1, -1):
            # mixing head of step t in one launch: the feedback gradient (x_{t+1} = m_t) joins through tanh',
            # then the two tiny Dense layers backwards; weight gradients are formed after the loop
            dh2_dense = ops.mix_head_bwd(dloss_tm[t], None if dx_next is None else dx_next.reshape(B, O), M[t], P[t], Wm_p_c,
                                         w["dense_W"], dpre_all[t], dpre_p_all[t])
            # layer 2 step: its input is h1_t, so its dx (= dz2 . K2^T) is the gradient w.r.t. h1_t
            b2 = ops.lstm_seq_bwd(H1[t + 1].view(B, 1, H), w["dec2_K"], w["dec2_R"], H2[t + 1].view(B, 1, H), R2[t],
                                  h0=H2[t], c0=C2[t], dhs=dh2_dense.reshape(B, 1, H), dhT=dh2_rec, dcT=dc2,
                                  need_dx=True, need_state_grads=True, act=act, dz=DZ2[t].view(B, 1, 4 * H), scratch=bsc,
                                  need_weight_grads=False)
            dh2_rec, dc2 = b2["dh0"], b2["dc0"]
            b1 = ops.lstm_seq_bwd(X[t].view(B, 1, O), w["dec1_K"], w["dec1_R"], H1[t + 1].view(B, 1, H), R1[t], h0=H1[t],
                                  c0=C1[t], dhs=b2["dx"], dhT=dh1_rec, dcT=dc1, need_dx=(t > 0), need_state_grads=True,
                                  act=act, dz=DZ1[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh1_rec, dc1 = b1["dh0"], b1["dc0"]
            dx_next = b1["dx"]
        # weight gradients of the unrolled decoder: x^T dz (input kernels, biases) and h_prev^T dz (recurrent)
        TB = T_out * B
        fl = lambda a, n: a.reshape(TB, n)
        # head: dense_W, dense_b, mix_W (others' rows and the prediction's), mix_b are adjacent in the flat buffer - one launch
        # and one reduce form all four, reading `others` in the (B,T,...) layout it arrived in
        # The decoder's weight-gradient products depend on nothing that follows, and what follows - the encoder's BPTT - are
        # latency-bound persistent kernels (one workgroup per CU, ~300 of 512 registers, matrix pipe a quarter busy) that leave
        # room for GEMM workgroups on the same CUs: the products go to a SIDE stream and run under them
        # (FOV_WGRAD_STREAM=0: in line).  The step's deferred reduce and the optimizer wait for the side stream.
        def decoder_wgrads(part):
            if part & 1:
                # head: dense_W, dense_b, mix_W (others' rows and the prediction's), mix_b are adjacent in the flat buffer - one
                # launch and one reduce form all four, reading `others` in the (B,T,...) layout it arrived in
                ops.mix_head_wgrad(H2[1:], dpre_p_all, others, P, dpre_all, self._span("dense_W", "mix_b"), accumulate=acc, scratch=sc)
                # a layer's kernel, recurrent kernel and bias are adjacent in the flat buffer: [h1_t | h2_{t-1} | 1]^T dz2 is ONE
                # product + one reduce (dz2 read once), [h1_{t-1} | 1]^T dz1 likewise; the 6-wide dK1 stays a skinny product
                ops.wgrad_fused(fl(H1[1:], H), fl(H2[:T_out], H), fl(DZ2, 4 * H), self._span("dec2_K", "dec2_b"), accumulate=acc,
                                scratch=sc, dtype=dt)
            if part & 2:
                ops.dense_bwd(fl(X, O), w["dec1_K"], fl(DZ1, 4 * H), dW=g["dec1_K"], need_db=False, need_dx=False,
                              accumulate=acc, scratch=sc)
                ops.wgrad_fused(fl(H1[:T_out], H), None, fl(DZ1, 4 * H), self._span("dec1_R", "dec1_b"), accumulate=acc, scratch=sc,
                                dtype=dt)
        side = self._wgrad_side_stream() if not (self._dp and self.overlap_allreduce) else None
        # the first step of a shape runs in line: the products' scratch buffer grows to its final size on THIS stream (a buffer
        # replaced while the side stream still reads the old one could be handed to another tensor)
        shape_key = (B, T_in, T_out, dt)
        if side is not None and shape_key not in self._side_warm:
            self._side_warm.add(shape_key)
            side = None
        main = torch.cuda.current_stream()
        # bf16: the products are short enough to be dealt under BOTH recurrences (half under layer 2's, half under layer 1's:
        # 0.506 -> 0.500 ms); fp32: all under layer 2's (dealt: 0.846 -> 0.859 ms).  FOV_WGRAD_SPLIT=0/1 overrides.
        split_side = side is not None and os.environ.get("FOV_WGRAD_SPLIT", "1" if dt == "bf16" else "0") == "1"
        if side is None:
            decoder_wgrads(3)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                decoder_wgrads(1 if split_side else 3)
                ops.reduce_defer_flush(self.grad)   # their slices are summed on the side stream too (under the recurrences), not at the step's end
        self.grads_final("dec1_K")    # decoder, heads and loss: all-reduced under the encoder's BPTT
        # encoder: layer 2 over hs1 (its dx is the dhs of layer 1), then layer 1
        # (tried: each encoder layer in two parts, its weight-gradient products on the side stream under the NEXT layer's
        # recurrence - 0.846 -> 0.881 ms: with the decoder's products already there the CUs' matrix pipes are full and the
        # recurrences slow down more than the products gain)
        # bf16: layer 2's own products go under layer 1's recurrence too (they sat between the two BPTT launches, 35 us of the
        # critical path); fp32: in line (its products are 3x as long, the recurrences slow down more than is gained)
        enc_side = side is not None and os.environ.get("FOV_WGRAD_ENC_SIDE", "1" if dt == "bf16" else "0") == "1"
        if enc_side:
            e2 = ops.lstm_seq_bwd(hs1, w["enc2_K"], w["enc2_R"], hs2, res2, dhT=dh2_rec, dcT=dc2, need_dx=True, act=act,
                                  scratch=bsc, dtype=dt, need_weight_grads=False)
        else:
            e2 = ops.lstm_seq_bwd(hs1, w["enc2_K"], w["enc2_R"], hs2, res2, dhT=dh2_rec, dcT=dc2, dK=g["enc2_K"], dR=g["enc2_R"],
                                  db=g["enc2_b"], need_dx=True, act=act, accumulate=acc, scratch=bsc, dtype=dt)
        if split_side or enc_side:   # the second half of the decoder's products under layer 1's recurrence
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if enc_side:
                    ops.lstm_seq_wgrad(hs1, hs2, e2["dz"], dK=g["enc2_K"], dR=g["enc2_R"], db=g["enc2_b"], accumulate=acc,
                                       scratch=self._side_scratch(), dtype=dt)
                if split_side:
                    decoder_wgrads(2)
                ops.reduce_defer_flush(self.grad)
        ops.lstm_seq_bwd(enc, w["enc1_K"], w["enc1_R"], hs1, res1, dhs=e2["dx"], dhT=dh1_rec, dcT=dc1, dK=g["enc1_K"],
                         dR=g["enc1_R"], db=g["enc1_b"], act=act, accumulate=acc, scratch=bsc, dtype=dt)
        if side is not None:
            main.wait_stream(side)
        return loss, out

    _side_stream = None
    _side_scratch_buf = None

    def _side_scratch(self):
        """Scratch of the encoder products that run on the side stream (the BPTT workspace is busy under them)."""
        if self._side_scratch_buf is None:
            self._side_scratch_buf = ops.Scratch()
        return self._side_scratch_buf

    @property
    def _side_warm(self):
        if "_side_warm_set" not in self.__dict__:
            self.__dict__["_side_warm_set"] = set()
        return self.__dict__["_side_warm_set"]

    def _wgrad_side_stream(self):
        # fp32: the products run under the encoder's recurrences on a second stream (0.818 -> 0.801 ms).  bf16: in line - its
        # recurrences take a whole CU's registers per workgroup and slow down by what a co-resident product gains, what is left
        # of the second stream are its hand-offs (r04, same box: 0.4557 ms with it, 0.4467 ms without).  FOV_WGRAD_STREAM=0/1 overrides.
        if os.environ.get("FOV_WGRAD_STREAM", "0" if getattr(self, "dtype", "f32") == "bf16" else "1") == "0":
            return None
        if self._side_stream is None:
            # LOW priority: when a product and a persistent recurrence kernel become ready together (both wait for the same
            # BPTT launch), the recurrence's workgroups are placed first and the product's fill what is left of every CU; at equal
            # priority the product's 512 blocks take the registers and the recurrence starts only when they have drained
            # (r04 timeline: 42 us between the two encoder BPTT launches).  FOV_SIDE_PRIORITY=normal: a plain torch stream.
            if os.environ.get("FOV_SIDE_PRIORITY", "low") == "low":
                self._side_stream = ops.side_stream(self.device, 1, owner=self)
            else:
                self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream


_CONV_MIX_ORDER = tuple(k for k in _MIX_ORDER if not k.startswith("mix_")) + \
    ("mixc0_W", "mixc0_b", "mixc1_W", "mixc1_b", "mixc2_W", "mixc2_b")


class OthersConvMixingTrainer(FlatParamTrainer):
    """Training step of the `conv_mixing` form of the others-mixing model
    (mycode/given_others_gt_mean_var_seq2seq.py:56,188-197,284-290,308): the mixing head is three Conv2D(1x3, same, relu)
    layers over the 1 x 6 map whose channels are the U users (others' mean/variance of the step, then the decoder's own
    prediction).  Step-wise on the layer kernels and the implicit-GEMM convolution; the data gradient of a convolution is
    the convolution with the transposed, flipped kernel; every weight gradient is ONE product over all steps."""

    def __init__(self, weights, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda"):
        self.act, self.impl = act, impl
        self._alloc(weights, _CONV_MIX_ORDER, optimizer, lr, device)

    def forward_backward(self, enc, others, dec0, target, grad_weight=1.0):
        w, g, act, impl, ws, sc, bsc = self.w, self.g, self.act, self.impl, self.ws, self.scratch, self.bwd_scratch
        B, T_in, _ = enc.shape
        T_out = others.shape[1]
        H, O = w["enc1_R"].shape[0], w["dense_W"].shape[1]
        U = w["mixc0_W"].shape[2]
        N0, N1 = w["mixc0_W"].shape[3], w["mixc1_W"].shape[3]
        e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
        self.grad.zero_()
        # ---------------- forward ----------------
        H1, C1, H2, C2 = e(T_out + 1, B, H), e(T_out + 1, B, H), e(T_out + 1, B, H), e(T_out + 1, B, H)
        hs1, _, _, res1 = ops.lstm_seq_train(enc, w["enc1_K"], w["enc1_R"], w["enc1_b"], act=act, impl=impl, workspace=ws,
                                             out=(e(B, T_in, H), H1[0], C1[0], e(B, T_in, 5, H)))
        hs2, _, _, res2 = ops.lstm_seq_train(hs1, w["enc2_K"], w["enc2_R"], w["enc2_b"], act=act, impl=impl, workspace=ws,
                                             out=(e(B, T_in, H), H2[0], C2[0], e(B, T_in, 5, H)))
        othT = others.permute(1, 0, 3, 2).contiguous()       # (T_out,B,6,U-1): the users as channels of a 1 x 6 map
        XM = e(T_out + 1, B, O)                               # row t = decoder input x_t, row t+1 = output m_t
        X, M = XM[:T_out], XM[1:]
        R1, R2 = e(T_out, B, 1, 5, H), e(T_out, B, 1, 5, H)
        P, A1, A2 = e(T_out, B, O), e(T_out, B, 1, O, N0), e(T_out, B, 1, O, N1)
        X[0].copy_(dec0.reshape(B, O))
        for t in range(T_out):
            ops.lstm_seq_train(X[t].view(B, 1, O), w["dec1_K"], w["dec1_R"], w["dec1_b"], H1[t], C1[t], act=act, impl=impl,
                               workspace=ws, out=(H1[t + 1].view(B, 1, H), None, C1[t + 1], R1[t]))
            ops.lstm_seq_train(H1[t + 1].view(B, 1, H), w["dec2_K"], w["dec2_R"], w["dec2_b"], H2[t], C2[t], act=act, impl=impl,
                               workspace=ws, out=(H2[t + 1].view(B, 1, H), None, C2[t + 1], R2[t]))
            ops.dense(H2[t + 1], w["dense_W"], w["dense_b"], activation="tanh", out=P[t])
            ops.conv2d_cat(othT[t].view(B, 1, O, U - 1), P[t].view(B, 1, O, 1), w["mixc0_W"], w["mixc0_b"], activation="relu", out=A1[t])
            ops.conv2d(A1[t], w["mixc1_W"], w["mixc1_b"], activation="relu", out=A2[t])
            ops.conv2d(A2[t], w["mixc2_W"], w["mixc2_b"], activation="relu", out=M[t].view(B, 1, O, 1))
        out = M.transpose(0, 1)
        # ---------------- backward ----------------
        dloss_tm, loss = ops.mse_dense_grad(M, target, None, scratch=sc, time_major=True)     # (T_out,B,O): dL/dm_t
        wt2, wt1 = ops.conv2d_weight_transpose(w["mixc2_W"]), ops.conv2d_weight_transpose(w["mixc1_W"])
        wt0_p = ops.conv2d_weight_transpose(w["mixc0_W"])[..., U - 1:].contiguous()           # only the prediction's channel
        D3, D2, D1 = e(T_out, B, 1, O, 1), e(T_out, B, 1, O, N1), e(T_out, B, 1, O, N0)       # pre-activation gradients
        DPRE, DZ1, DZ2 = e(T_out, B, O), e(T_out, B, 4 * H), e(T_out, B, 4 * H)
        dh1_rec = dc1 = dh2_rec = dc2 = dx_next = None
        for t in range(T_out - 1, -1, -1):
            m4 = M[t].view(B, 1, O, 1)
            if dx_next is None:
                ops.act_bwd(dloss_tm[t].view(B, 1, O, 1), m4, activation="relu", out=D3[t])
            else:        # x_{t+1} = m_t: the feedback gradient joins before the relu
                dy = ops.act_bwd(dx_next.reshape(B, O), M[t], base=dloss_tm[t], activation=None)
                ops.act_bwd(dy.view(B, 1, O, 1), m4, activation="relu", out=D3[t])
            ops.act_bwd(ops.conv2d(D3[t], wt2), A2[t], activation="relu", out=D2[t])
            ops.act_bwd(ops.conv2d(D2[t], wt1), A1[t], activation="relu", out=D1[t])
            dp = ops.conv2d(D1[t], wt0_p)                                                      # (B,1,O,1): dL/dp_t
            ops.act_bwd(dp.view(B, O), P[t], activation="tanh", out=DPRE[t])
            dh2_dense, _, _ = ops.dense_bwd(H2[t + 1], w["dense_W"], DPRE[t], need_dx=True, need_dW=False, need_db=False, scratch=sc)
            b2 = ops.lstm_seq_bwd(H1[t + 1].view(B, 1, H), w["dec2_K"], w["dec2_R"], H2[t + 1].view(B, 1, H), R2[t], h0=H2[t],
                                  c0=C2[t], dhs=dh2_dense.reshape(B, 1, H), dhT=dh2_rec, dcT=dc2, need_dx=True, need_state_grads=True,
                                  act=act, dz=DZ2[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh2_rec, dc2 = b2["dh0"], b2["dc0"]
            b1 = ops.lstm_seq_bwd(X[t].view(B, 1, O), w["dec1_K"], w["dec1_R"], H1[t + 1].view(B, 1, H), R1[t], h0=H1[t], c0=C1[t],
                                  dhs=b2["dx"], dhT=dh1_rec, dcT=dc1, need_dx=(t > 0), need_state_grads=True, act=act,
                                  dz=DZ1[t].view(B, 1, 4 * H), scratch=bsc, need_weight_grads=False)
            dh1_rec, dc1, dx_next = b1["dh0"], b1["dc0"], b1["dx"]
        TB = T_out * B
        fl = lambda a, n: a.reshape(TB, n)
        # mixing convolutions: one weight-gradient product per layer over all steps; the first layer's input is two maps
        ops.conv2d_wgrad(A2.view(TB, 1, O, N1), D3.view(TB, 1, O, 1), 1, 3, dw=g["mixc2_W"], scratch=sc)
        ops.conv2d_wgrad(A1.view(TB, 1, O, N0), D2.view(TB, 1, O, N1), 1, 3, dw=g["mixc1_W"], scratch=sc)
        g0_o = ops.conv2d_wgrad(othT.view(TB, 1, O, U - 1), D1.view(TB, 1, O, N0), 1, 3, scratch=sc)
        g0_p = ops.conv2d_wgrad(P.view(TB, 1, O, 1), D1.view(TB, 1, O, N0), 1, 3, scratch=sc)
        g["mixc0_W"][:, :, :U - 1].copy_(g0_o)
        g["mixc0_W"][:, :, U - 1:].copy_(g0_p)
        for i, D in enumerate((D1, D2, D3)):
            ops.colsum(D.view(TB * O, -1), out=g["mixc%d_b" % i], scratch=sc)
        ops.dense_bwd(fl(H2[1:], H), w["dense_W"], fl(DPRE, O), dW=g["dense_W"], db=g["dense_b"], need_dx=False, scratch=sc)
        ops.dense_bwd(fl(H1[1:], H), w["dec2_K"], fl(DZ2, 4 * H), dW=g["dec2_K"], db=g["dec2_b"], need_dx=False, scratch=sc)
        ops.dense_bwd(fl(H2[:T_out], H), w["dec2_R"], fl(DZ2, 4 * H), dW=g["dec2_R"], need_db=False, need_dx=False, scratch=sc)
        ops.dense_bwd(fl(X, O), w["dec1_K"], fl(DZ1, 4 * H), dW=g["dec1_K"], db=g["dec1_b"], need_dx=False, scratch=sc)
        ops.dense_bwd(fl(H1[:T_out], H), w["dec1_R"], fl(DZ1, 4 * H), dW=g["dec1_R"], need_db=False, need_dx=False, scratch=sc)
        e2 = ops.lstm_seq_bwd(hs1, w["enc2_K"], w["enc2_R"], hs2, res2, dhT=dh2_rec, dcT=dc2, dK=g["enc2_K"], dR=g["enc2_R"],
                              db=g["enc2_b"], need_dx=True, act=act, scratch=bsc)
        ops.lstm_seq_bwd(enc, w["enc1_K"], w["enc1_R"], hs1, res1, dhs=e2["dx"], dhT=dh1_rec, dcT=dc1, dK=g["enc1_K"],
                         dR=g["enc1_R"], db=g["enc1_b"], act=act, scratch=bsc)
        loss = self._weigh(loss, grad_weight)
        return loss, out


def convlstm_weight_order(weights):
    """Parameter order of the ConvLSTM seq2seq: enc0..2, dec0..2 (K, R, b each), then the head layers."""
    order = []
    for side in ("enc", "dec"):
        for l in range(3):
            order += ["%s%d_K" % (side, l), "%s%d_R" % (side, l), "%s%d_b" % (side, l)]
    i = 0
    while "head%d_W" % i in weights:
        order += ["head%d_W" % i, "head%d_b" % i]
        i += 1
    return order


class ConvLSTMTrainer(FlatParamTrainer):
    """Training step of the ConvLSTM2D seq2seq (mycode/convlstm_seq2seq.py:100-287: 3-layer encoder, 3-layer
    decoder unrolled `predict_step` times with its own prediction fed back, conv head + channel softmax,
    `model.compile(optimizer='RMSprop', loss=costfunc._mse)`).

    Forward keeps a time-major tape (every map of a step is batch-dense, steps stack into one pixel axis):
    layer outputs, activated gates, cell states, head activations.  Backward walks the unrolled decoder in
    reverse (softmax -> head -> three cells; the feedback path's gradient is the dx of decoder layer 0), then
    the encoder layer by layer; the recurrent data gradients are forward convolutions with transposed
    weights, and every weight / bias gradient is ONE product over all steps of a layer (conv2d_wgrad over the
    stacked steps).  All arithmetic is in libfov360_hip.so; torch holds the buffers (and adds maps).

    Input dropout (cfg.dropout_rate, Keras ConvLSTM2D `dropout=`, convlstm_seq2seq.py:103,111,121,149,156,163):
    Keras draws FOUR masks per layer call (one per gate i,f,c,o; inverted dropout, 1/(1-rate) on the kept
    entries), reuses them over the time steps of that call, and convolves each masked copy of the input with
    its gate's slice of the kernel.  The encoder layers are called once per batch, the decoder layers once
    per unrolled step.  Here the four masked copies are stacked along the channel axis and convolved with the
    block-diagonal (kh,kw,4C,4F) arrangement of the kernel - the same arithmetic on the same conv kernels
    (4x the input-convolution FLOPs, training with dropout only).  Masks come from a torch generator (`seed`);
    TF's random stream is not reproducible, the distribution is the same.

    dilation_rate (cfg.dilation_rate, convlstm_seq2seq.py:102,110,120,148,155,162): Keras's ConvLSTM2DCell passes it to the
    input convolution only (`input_conv(..., dilation_rate=self.dilation_rate)`; `recurrent_conv` has none), so K's taps are
    spread, R's are not; the head's Conv2D layers are not dilated.  Backward: dx through the dilated convolution with the
    transposed K, dK through the dilated weight gradient."""

    def __init__(self, weights, head="conv2d", act="hard_sigmoid", optimizer="rmsprop", lr=1e-3, device="cuda",
                 dropout_rate=0.0, seed=0, add_xyz_sum1=False, loss="mse", dilation_rate=1):
        if not 0.0 <= dropout_rate < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        if int(dilation_rate) < 1:
            raise ValueError("dilation_rate must be >= 1")
        self.dilation = int(dilation_rate)          # cfg.dilation_rate (config.py:105): the six ConvLSTM2D INPUT convolutions
        self.dropout_rate = float(dropout_rate)
        self.add_xyz_sum1 = bool(add_xyz_sum1)      # cfg.add_xyz_sum1: the optional unit-norm term of costfunc._mse
        if loss not in ("mse", "categorical_crossentropy"):
            raise ValueError("loss must be 'mse' or 'categorical_crossentropy'")
        self.loss = loss
        self._gen = torch.Generator(device=device)
        self._gen.manual_seed(int(seed))
        self.head, self.act = head, act
        self.order = convlstm_weight_order(weights)
        self._alloc(weights, self.order, optimizer, lr, device)   # ws / bwd_scratch stay unused; the shared fit loop checks them
        self.n_head = sum(1 for k in self.order if k.startswith("head") and k.endswith("_W"))
        self.filters = [self.w["enc%d_R" % l].shape[2] for l in range(3)]

    # -- forward with tape ---------------------------------------------------------------------------
    def sample_masks(self, B, H, W, C, T_out):
        """Dropout masks of one training step: 'enc{l}' (4,B,H,W,C_l), 'dec{l}' (T_out,4,B,H,W,C_l)."""
        keep = 1.0 - self.dropout_rate
        cin = [C] + self.filters[:2]
        m = {}
        for l in range(3):
            for key, lead in (("enc%d" % l, (4,)), ("dec%d" % l, (T_out, 4))):
                u = torch.rand(lead + (B, H, W, cin[l]), generator=self._gen, device=self.device)
                m[key] = (u < keep).to(torch.float32) / keep
        return m

    @staticmethod
    def _stack_masked(x, mask4):
        """x (..., C) and mask4 (4, ..., C) -> (..., 4C): the four masked copies side by side, gate-major."""
        return torch.cat([x * mask4[g] for g in range(4)], dim=-1)

    @staticmethod
    def _block_diag_kernel(K):
        """(kh,kw,C,4F) -> (kh,kw,4C,4F): gate g's kernel slice sees only the g-th masked copy of the input."""
        kh, kw, C, F4 = K.shape
        F = F4 // 4
        K4 = torch.zeros((kh, kw, 4 * C, F4), dtype=K.dtype, device=K.device)
        for g in range(4):
            K4[:, :, g * C:(g + 1) * C, g * F:(g + 1) * F] = K[..., g * F:(g + 1) * F]
        return K4

    def _forward(self, enc, dec0, T_out, masks=None):
        w, act, F, dil = self.w, self.act, self.filters, self.dilation
        B, T_in, H, W, C = enc.shape
        dev = enc.device
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        tape = {"x": enc.permute(1, 0, 2, 3, 4).contiguous()}           # (T_in,B,H,W,C) time-major
        k4 = {} if masks is None else {k: self._block_diag_kernel(w[k]) for k in self.order if k.endswith("_K")}
        tape["masks"], tape["k4"] = masks, k4
        seq = tape["x"]
        for l in range(3):
            K, R, b = w["enc%d_K" % l], w["enc%d_R" % l], w["enc%d_b" % l]
            hs, cs, gs = e(T_in, B, H, W, F[l]), e(T_in, B, H, W, F[l]), e(T_in, B, H, W, 4 * F[l])
            if masks is not None:   # one mask set for the whole sequence of this layer call
                seq = self._stack_masked(seq, masks["enc%d" % l].unsqueeze(1))
                K = k4["enc%d_K" % l]
                tape["ex4_%d" % l] = seq
            KR = torch.cat([K, R], 2)     # [K ; R]: input and recurrent convolution of a step in one launch
            for t in range(T_in):
                if t > 0:     # the whole step (both convolutions, gates, c / h update, gates tape) is one launch
                    ops.convlstm_cell(seq[t], hs[t - 1], KR, b, cs[t - 1], hs[t], act, c_new=cs[t], gates=gs[t], dilation=dil)
                else:         # zero initial state
                    ops.convlstm_cell(seq[t], None, K, b, None, hs[t], act, c_new=cs[t], gates=gs[t], dilation=dil)
            tape["eh%d" % l], tape["ec%d" % l], tape["eg%d" % l] = hs, cs, gs
            seq = hs
        cat = sum(F)
        offs = [0, F[0], F[0] + F[1]]
        dense_head = self.head == "dense"
        Co = 6 if dense_head else w["head%d_W" % (self.n_head - 1)].shape[3]
        feat = e(T_out, B, H, W, cat)
        inp = e(T_out, B, H, W, C)                                      # decoder inputs, step by step
        inp[0].copy_(dec0[:, 0])
        P = e(T_out, B, Co) if dense_head else e(T_out, B, H, W, Co)
        dcs = [e(T_out, B, H, W, F[l]) for l in range(3)]
        dgs = [e(T_out, B, H, W, 4 * F[l]) for l in range(3)]
        ys = [] if dense_head else [e(T_out, B, H, W, w["head%d_W" % i].shape[3]) for i in range(self.n_head)]
        last_act = "relu" if self.head == "conv2d" else None
        cin = [C] + F[:2]
        dx4 = [e(T_out, B, H, W, 4 * cin[l]) for l in range(3)] if masks is not None else None
        kr = [torch.cat([k4["dec%d_K" % l] if masks is not None else w["dec%d_K" % l], w["dec%d_R" % l]], 2) for l in range(3)]
        for t in range(T_out):
            cur = inp[t]
            for l in range(3):
                K, R, b = w["dec%d_K" % l], w["dec%d_R" % l], w["dec%d_b" % l]
                h_prev = tape["eh%d" % l][T_in - 1] if t == 0 else feat[t - 1][..., offs[l]:offs[l] + F[l]]
                c_prev = tape["ec%d" % l][T_in - 1] if t == 0 else dcs[l][t - 1]
                if masks is not None:   # a fresh mask set for every unrolled call of the decoder layer
                    dx4[l][t].copy_(self._stack_masked(cur, masks["dec%d" % l][t]))
                    cur, K = dx4[l][t], k4["dec%d_K" % l]
                hslot = feat[t][..., offs[l]:offs[l] + F[l]]
                ops.convlstm_cell(cur, h_prev, kr[l], b, c_prev, hslot, act, c_new=dcs[l][t], gates=dgs[l][t], dilation=dil)
                cur = hslot
            if dense_head:     # Flatten + Dense(6, linear); fed back as a 1x1x6 map
                ops.dense(feat[t].reshape(B, H * W * cat), w["head0_W"], w["head0_b"], activation=None, out=P[t])
            else:
                y = feat[t]
                for i in range(self.n_head):
                    a = "relu" if i < self.n_head - 1 else last_act
                    y = ops.conv2d(y, w["head%d_W" % i], w["head%d_b" % i], activation=a, out=ys[i][t])
                ops.softmax_lastdim(y, out=P[t])
            if t + 1 < T_out:
                assert P[t].numel() == inp[t + 1].numel(), "the head's output is fed back as the next decoder input"
                inp[t + 1].copy_(P[t].reshape(inp[t + 1].shape))
        tape.update(feat=feat, inp=inp, P=P, dc=dcs, dg=dgs, ys=ys, offs=offs, last_act=last_act, dx4=dx4)
        return P, tape

    def forward_backward(self, enc, dec0, target, grad_weight=1.0, masks=None):
        """enc (B,T_in,H,W,C), dec0 (B,1,H,W,C), target (B,T_out,H,W,Co) device tensors.  Fills self.grad with
        d(mean squared error)/d(parameters); returns (loss (1,), prediction (B,T_out,H,W,Co)).  `masks`: dropout
        masks as sample_masks() returns them (drawn here when dropout_rate > 0 and none are given)."""
        w, g, act, F, sc, dil = self.w, self.g, self.act, self.filters, self.scratch, self.dilation
        B, T_in, H, W, C = enc.shape
        T_out = target.shape[1]
        if masks is None and self.dropout_rate > 0:
            masks = self.sample_masks(B, H, W, C, T_out)
        P, tp = self._forward(enc, dec0, T_out, masks)
        offs, feat = tp["offs"], tp["feat"]
        cin = [C] + F[:2]

        def unmask(d4, mask4, c):      # data gradient of the stacked input -> gradient of the unmasked map
            out = d4[..., :c] * mask4[0]
            for gi in range(1, 4):
                out = out + d4[..., gi * c:(gi + 1) * c] * mask4[gi]
            return out

        def in_kernel_grad(key, x4, dz, c, f):   # weight gradient of the block-diagonal kernel -> its diagonal blocks
            kh, kw = w[key].shape[:2]
            d4 = ops.conv2d_wgrad(x4, dz, kh, kw, scratch=sc, dilation=dil)
            for gi in range(4):
                g[key][..., gi * f:(gi + 1) * f].copy_(d4[:, :, gi * c:(gi + 1) * c, gi * f:(gi + 1) * f])

        wt4 = {k: ops.conv2d_weight_transpose(v) for k, v in tp["k4"].items()}
        dense_head = self.head == "dense"
        tgt = target.transpose(0, 1).contiguous()
        if self.loss == "categorical_crossentropy":      # convlstm_heatmap.py:192
            dP, loss = ops.categorical_crossentropy_grad(P, tgt, scratch=sc)
        else:
            dP, loss = ops.mse_dense_grad(P, tgt, None, scratch=sc)
        if self.add_xyz_sum1:      # costfunc._mse, cost.py:23-28: + 0.5 * MSE(1, ux^2 + uy^2 + uz^2)
            reg = ops.xyz_sum1_grad(P, dP, scratch=sc)
            loss = ops.act_bwd(reg, reg, base=loss, activation=None)
        wt = {k: ops.conv2d_weight_transpose(w[k]) for k in self.order if k.endswith(("_K", "_R")) or
              (k.endswith("_W") and not dense_head)}
        nh = self.n_head
        dys = [torch.empty_like(y) for y in tp["ys"]]                    # d(pre-activation) of every head layer
        dzs = [torch.empty_like(x) for x in tp["dg"]]                    # decoder dz, all steps
        zeros = lambda l: torch.zeros((B, H, W, F[l]), dtype=torch.float32, device=enc.device)
        dh_rec, dc = [zeros(l) for l in range(3)], [zeros(l) for l in range(3)]
        dfeed = None
        for t in range(T_out - 1, -1, -1):
            dp = dP[t] if dfeed is None else dP[t] + dfeed.reshape(dP[t].shape)
            if dense_head:
                if dfeed is not None:
                    dP[t].copy_(dp)                                      # kept: the Dense weight gradient is formed below
                dfeat, _, _ = ops.dense_bwd(feat[t].reshape(B, -1), w["head0_W"], dP[t], need_dW=False, need_db=False, scratch=sc)
                dfeat = dfeat.reshape(B, H, W, -1)
            else:
                d = ops.softmax_lastdim_bwd(dp, P[t], out=dys[nh - 1][t])
                if tp["last_act"] == "relu":
                    ops.act_bwd(d, tp["ys"][nh - 1][t], activation="relu", out=d)
                for i in range(nh - 1, 0, -1):                           # head layers, data gradient only
                    d = ops.conv2d(d, wt["head%d_W" % i], out=dys[i - 1][t])
                    ops.act_bwd(d, tp["ys"][i - 1][t], activation="relu", out=d)
                dfeat = ops.conv2d(d, wt["head0_W"])
            dx_up = None
            for l in range(2, -1, -1):
                dh = dfeat[..., offs[l]:offs[l] + F[l]] + dh_rec[l]
                if dx_up is not None:
                    dh = dh + dx_up
                c_prev = tp["ec%d" % l][T_in - 1] if t == 0 else tp["dc"][l][t - 1]
                dz = ops.convlstm_gates_bwd(dh, dc[l], tp["dg"][l][t], c_prev, tp["dc"][l][t], act, dz=dzs[l][t])
                dh_rec[l] = ops.conv2d(dz, wt["dec%d_R" % l])
                if l > 0 or t > 0:
                    if masks is None:
                        dx = ops.conv2d(dz, wt["dec%d_K" % l], dilation=dil)
                    else:
                        dx = unmask(ops.conv2d(dz, wt4["dec%d_K" % l], dilation=dil), masks["dec%d" % l][t], cin[l])
                    if l > 0:
                        dx_up = dx
                    else:
                        dfeed = dx
        # head weight gradients: one product over all steps per layer
        x_in = feat
        if dense_head:
            ops.dense_bwd(feat.reshape(T_out * B, -1), w["head0_W"], dP.reshape(T_out * B, -1), dW=g["head0_W"], db=g["head0_b"],
                          need_dx=False, scratch=sc)
        for i in range(0 if dense_head else nh):
            ops.conv2d_wgrad(x_in, dys[i], *w["head%d_W" % i].shape[:2], dw=g["head%d_W" % i], scratch=sc)
            ops.colsum(dys[i], out=g["head%d_b" % i], scratch=sc)
            x_in = tp["ys"][i]
        # decoder cells
        for l in range(3):
            x_in = tp["inp"] if l == 0 else feat[..., offs[l - 1]:offs[l - 1] + F[l - 1]]
            kh, kw = w["dec%d_K" % l].shape[:2]
            if masks is None:
                ops.conv2d_wgrad(x_in, dzs[l], kh, kw, dw=g["dec%d_K" % l], scratch=sc, dilation=dil)
            else:
                in_kernel_grad("dec%d_K" % l, tp["dx4"][l], dzs[l], cin[l], F[l])
            ops.conv2d_wgrad(tp["eh%d" % l][T_in - 1], dzs[l][0], kh, kw, dw=g["dec%d_R" % l], scratch=sc)
            if T_out > 1:
                ops.conv2d_wgrad(feat[:T_out - 1][..., offs[l]:offs[l] + F[l]], dzs[l][1:], kh, kw, dw=g["dec%d_R" % l],
                                 accumulate=True, scratch=sc)
            ops.colsum(dzs[l], out=g["dec%d_b" % l], scratch=sc)
        # encoder, top layer first; dh_rec / dc arrive from the decoder's first step
        dx_seq = None
        for l in range(2, -1, -1):
            gs, cs, hs = tp["eg%d" % l], tp["ec%d" % l], tp["eh%d" % l]
            edz = torch.empty_like(gs)
            dhr, dcl = dh_rec[l], dc[l]
            for t in range(T_in - 1, -1, -1):
                dh = dhr if dx_seq is None else dhr + dx_seq[t]
                dz = ops.convlstm_gates_bwd(dh, dcl, gs[t], cs[t - 1] if t > 0 else None, cs[t], act, dz=edz[t])
                if t > 0:
                    dhr = ops.conv2d(dz, wt["enc%d_R" % l])
            kh, kw = w["enc%d_K" % l].shape[:2]
            x_in = tp["x"] if l == 0 else tp["eh%d" % (l - 1)]
            if masks is None:
                ops.conv2d_wgrad(x_in, edz, kh, kw, dw=g["enc%d_K" % l], scratch=sc, dilation=dil)
            else:
                in_kernel_grad("enc%d_K" % l, tp["ex4_%d" % l], edz, cin[l], F[l])
            if T_in > 1:
                ops.conv2d_wgrad(hs[:T_in - 1], edz[1:], kh, kw, dw=g["enc%d_R" % l], scratch=sc)
            else:
                g["enc%d_R" % l].zero_()
            ops.colsum(edz, out=g["enc%d_b" % l], scratch=sc)
            if l > 0:   # data gradient for the layer below, all steps in one launch
                if masks is None:
                    dx_seq = ops.conv2d(edz.reshape(T_in * B, H, W, 4 * F[l]), wt["enc%d_K" % l], dilation=dil).reshape(T_in, B, H, W, F[l - 1])
                else:
                    d4 = ops.conv2d(edz.reshape(T_in * B, H, W, 4 * F[l]), wt4["enc%d_K" % l], dilation=dil).reshape(T_in, B, H, W, 4 * F[l - 1])
                    dx_seq = unmask(d4, masks["enc%d" % l].unsqueeze(1), F[l - 1])
        loss = self._weigh(loss, grad_weight)
        return loss, P.transpose(0, 1)

    def eval_loss(self, enc, dec0, target):
        P, _ = self._forward(enc, dec0, target.shape[1])
        tgt = target.transpose(0, 1).contiguous()
        if self.loss == "categorical_crossentropy":
            dP, loss = ops.categorical_crossentropy_grad(P, tgt, scratch=self.scratch)
        else:
            dP, loss = ops.mse_dense_grad(P, tgt, None, scratch=self.scratch)
        if self.add_xyz_sum1:
            reg = ops.xyz_sum1_grad(P, dP, scratch=self.scratch)
            loss = ops.act_bwd(reg, reg, base=loss, activation=None)
        return loss


class TFLSTMTrainer(FlatParamTrainer):
    """Training step of the raw-TensorFlow model of mycode/lstm.py (cfg.use_xyz, cfg.predict_mean_var,
    predict_len == 1 form): MultiRNNCell of LSTMCell(n_hidden) under dynamic_rnn with a fed state (:218-240),
    the two two-layer heads of _pred_mean_var_xyz2_new on the top layer's final h (:321-337: relu -> tanh for the
    means, relu -> linear -> exp for the variances), costfunc.likelihood_loss_tf (cost.py:190-229) and
    tf.train.RMSPropOptimizer with the script's clip_by_value(-1, 1) (lstm.py:556-567).

    `cells`: [(W (F+H,4H), b (4H))] in tf.contrib LSTMCell layout; they are trained in the kernels' Keras layout
    (a column permutation plus the constant forget_bias: an element-wise optimizer does not see the difference) and
    converted back by `cells_tf()`.  `head`: dict mu_W1 (H,32), mu_b1, mu_W2 (32,3), mu_b2, var_W1, var_b1, var_W2,
    var_b2.  The DropoutWrapper (output_keep_prob) acts on what a layer hands UP, not on its recurrent state: give
    `masks` [(B,T,H) per non-top layer, already scaled by 1/keep] to reproduce it; the head reads the top state h,
    which no mask touches.  The predict_len > 1 form of the script (the value mycode/config.py:21 ships) re-feeds
    sampled seconds: pass `noise` to forward_backward / train_step (fov_sample_refeed_fwd / _bwd).

    A FlatParamTrainer like the Keras-path trainers: parameters / gradients / the RMSProp slot are flat buffers, check() reports a
    persistent kernel that gave up, the optimizer launch is guarded by the workspaces' timeout words (skipped on the device,
    `applied` counts the updates that ran), and under torch.distributed every rank passes its shard and `n_global`: ONE SUM
    all-reduce of [poison | gradients | loss] per step, the clip and the RMSProp update act on the all-reduced gradient as the
    single-process step does on the global batch's.  The epoch loop, learning-rate schedule, tf.train.Saver-style save /
    restore and the test loop of lstm.py:583-828 are in lstm_driver.py."""

    HEAD = ("mu_W1", "mu_b1", "mu_W2", "mu_b2", "var_W1", "var_b1", "var_W2", "var_b2")
    GMM_HEAD = ("fc1_W", "fc1_b", "fc2_W", "fc2_b", "fc3_W", "fc3_b", "fc4_W", "fc4_b")
    RAW_HEAD = ("conv1_W", "conv1_b", "conv2_W", "conv2_b", "conv3_W", "conv3_b")
    HEADS = {"meanvar": HEAD, "gmm": GMM_HEAD, "raw": RAW_HEAD}
    # the one head tensor that reads h (its H rows are zero-padded to Hp): name -> axis of H
    FIRST = {"meanvar": {"mu_W1": 0, "var_W1": 0}, "gmm": {"fc1_W": 0}, "raw": {"conv1_W": 1}}

    @staticmethod
    def head_kind_of(cfg):
        """The branch of mycode/lstm.py:424-509 a configuration selects (cfg.use_xyz): predict_mean_var -> 'meanvar' (:426),
        elif use_GMM -> 'gmm' (:482 - what the committed config.py:69,71 runs), else 'raw' (:487)."""
        if not cfg.get("use_xyz", True):
            raise NotImplementedError("cfg.use_phi_theta branches of lstm.py (:511-527) are not on the hot path")
        if cfg.get("predict_mean_var", False):
            return "meanvar"
        return "gmm" if cfg.get("use_GMM", True) else "raw"

    @classmethod
    def from_cfg(cls, cfg, cells, head, **kw):
        """Trainer for the branch `cfg` selects, with the script's own hyper-parameters (lstm.py:59,554-567, config.py)."""
        kw.setdefault("lr", cfg.get("LEARNING_RATE", 1e-5))
        kw.setdefault("clip_value", 1.0 if cfg.get("clip_gradient", True) else 0.0)
        kw.setdefault("fps", cfg.get("fps", 30))
        kw.setdefault("running_length", cfg.get("running_length", 10))
        kw.setdefault("process_in_seconds", cfg.get("process_in_seconds", True))
        kw.setdefault("batch_size", cfg.get("batch_size", None))
        return cls(cells, head, head_kind=cls.head_kind_of(cfg), **kw)

    def __init__(self, cells, head, forget_bias=1.0, lr=1e-5, clip_value=1.0, decay=0.9, eps=1e-10, fps=30, running_length=10,
                 impl="auto", device="cuda", pad=True, head_kind="meanvar", n_mix=20, weight_by_pi=False, use_reg=False,
                 process_in_seconds=True, batch_size=None):
        from .models import convert_tf_lstmcell, padded_width, pad_lstm, MFMA_WIDTHS
        self.forget_bias, self.lr, self.clip, self.decay, self.eps = float(forget_bias), float(lr), float(clip_value), decay, eps
        self.fps, self.running_length, self.impl, self.device = fps, running_length, impl, device
        # head_kind: 'meanvar' (_pred_mean_var_xyz2_new + likelihood_loss_tf), 'gmm' (_GMM_3dgassian + mixture_3d_gaussian_loss,
        # lstm.py:377-400,482-485; cost.py:486-549) or 'raw' (pred_cnn_model_fn + MSE / pred_raw_loss_tf, lstm.py:147-174,486-508)
        self.head_kind, self.n_mix, self.weight_by_pi, self.use_reg = head_kind, int(n_mix), bool(weight_by_pi), bool(use_reg)
        self.process_in_seconds, self.batch_size = bool(process_in_seconds), batch_size
        self.HEAD = self.HEADS[head_kind]
        first = self.FIRST[head_kind]
        conv = [convert_tf_lstmcell(W, b, forget_bias) for W, b in cells]
        self.L = len(conv)
        # The stack runs at the next matrix-core width Hp (lstm.py's n_hidden = 400 -> 512: lstm_wide16.hip forward, lstm_bwd16.hip
        # BPTT) with zero-padded weights - exact, as in PaddedTrainer: a padded unit has z = 0, c = h = 0, dz = 0, so every
        # gradient in a padded slice is exactly 0 and TF's RMSProp leaves the zeros in place.  States, masks, weights and
        # gradients cross the object's surface at the caller's width H (cells_tf, grads_numpy, predict, train_step).
        self.H = int(conv[0][1].shape[0])
        Hp = padded_width(self.H, MFMA_WIDTHS + (512,)) if (pad and impl != "generic") else None
        self.Hp = int(Hp) if Hp else self.H
        if self.Hp != self.H:
            conv = [pad_lstm(K, R, b, self.Hp, pad_input=(l > 0)) for l, (K, R, b) in enumerate(conv)]
        weights = {}
        for l, (K, R, b) in enumerate(conv):
            weights.update({"K%d" % l: K, "R%d" % l: R, "b%d" % l: b})
        weights.update({k: np.ascontiguousarray(head[k], dtype=np.float32) for k in self.HEAD})
        if self.Hp != self.H:
            for k, ax in first.items():
                widths = [(0, 0)] * weights[k].ndim
                widths[ax] = (0, self.Hp - self.H)
                weights[k] = np.ascontiguousarray(np.pad(weights[k], widths))
        self.order = ["%s%d" % (n, l) for l in range(self.L) for n in ("K", "R", "b")] + list(self.HEAD)
        self._alloc(weights, self.order, "rmsprop_tf", lr, device)     # flat buffers, the [poison | grads | loss] layout, workspaces
        self.m.fill_(1.0)      # TF initialises the rms slot to one
        self.ms = self.m
        self._state_pads = {}
        self._carry = {}      # batch -> two (L,2,B,Hp) buffers the final state alternates between (train_step(..., state_view=True))

    def apply_gradients(self):
        """tf.train.RMSPropOptimizer on clip_by_value(grad, -1, 1) (lstm.py:556-567), one guarded launch on the flat buffers."""
        self.step_count += 1
        ops.rmsprop_tf_step(self.flat, self.grad, self.ms, self.lr, self.decay, self.eps, self.clip, guards=self._guards(),
                            applied=self.applied)

    def _unpadded(self, table, k):
        """Tensor `k` of self.w / self.g at the caller's width H (numpy)."""
        v = table[k].detach().cpu().numpy()
        H, Hp = self.H, self.Hp
        if Hp == H:
            return v
        if k[0] in "KRb" and k[1:].isdigit():
            if k[0] == "R" or (k[0] == "K" and int(k[1:]) > 0):
                v = v[:H]
            return _unpad_gates(v, H, Hp)
        ax = self.FIRST[self.head_kind].get(k)
        return v if ax is None else np.ascontiguousarray(np.take(v, np.arange(H), axis=ax))

    def weights_numpy(self):
        return {k: self._unpadded(self.w, k) for k in self.order}

    def grads_numpy(self):
        """Gradients of the last forward_backward, Keras layout, at width H."""
        return {k: self._unpadded(self.g, k) for k in self.order}

    def padded_slices_are_zero(self):
        """True if every padded slice of the parameters and of the last gradients is exactly zero (synchronises)."""
        if self.Hp == self.H:
            return True
        nz = 0
        for table in (self.w, self.g):
            for k in self.order:
                full = table[k].detach().cpu().numpy()
                nz += int(np.count_nonzero(full)) - int(np.count_nonzero(self._unpadded(table, k)))
        return nz == 0

    def _pad_state(self, st):
        """(L,2,B,H) -> (L,2,B,Hp), zero columns."""
        if st is None or self.Hp == self.H or st.shape[-1] == self.Hp:
            return st
        for buf in self._carry.get(st.shape[2], ()):      # the width-H view of a carried-state buffer this trainer returned: no copy
            if st.data_ptr() == buf.data_ptr() and st.stride() == buf.stride():
                return buf
        key = tuple(st.shape)
        out = self._state_pads.get(key)      # one buffer per shape: its padded columns are written once (zero), one copy per call
        if out is None:
            out = self._state_pads[key] = torch.zeros(st.shape[:-1] + (self.Hp,), dtype=torch.float32, device=st.device)
        out[..., :self.H].copy_(st)
        return out

    def _state_out(self, states):
        """[(cT, hT) per layer] at width Hp -> (L,2,B,H) in one copy."""
        H = self.H
        flat = [t[:, :H] for pair in states for t in pair]
        return torch.stack(flat, 0).view(len(states), 2, flat[0].shape[0], H)

    def _carry_target(self, B, init_state):
        """The carried-state buffer of batch B that does NOT hold `init_state` (the kernels read one and write the other)."""
        bufs = self._carry.get(B)
        if bufs is None:
            bufs = self._carry[B] = [torch.zeros((self.L, 2, B, self.Hp), dtype=torch.float32, device=self.device) for _ in range(2)]
        if init_state is not None and init_state.data_ptr() == bufs[0].data_ptr():
            return bufs[1]
        return bufs[0]

    def _pad_masks(self, masks):
        if masks is None or self.Hp == self.H:
            return masks
        out = []
        for m in masks:
            if m is None or m.shape[-1] == self.Hp:
                out.append(m)
            else:
                mp = torch.zeros(m.shape[:-1] + (self.Hp,), dtype=torch.float32, device=m.device)
                mp[..., :self.H].copy_(m)
                out.append(mp)
        return out

    def cells_tf(self):
        """Current LSTM weights back in tf.contrib LSTMCell layout [(W (F+H,4H), b)]."""
        return self._cells_from(self.w, True)

    def _head_fused(self, h):
        M, O = self.w["mu_W2"].shape
        return os.environ.get("FOV_NO_TF_HEAD") is None and ops.tf_head_supported(h.shape[0], h.shape[1], M, O)

    def _mlp_layers(self, table):
        """[(W, b, activation)] of the fused chain (mlp_head.hip) over `table` = self.w or self.g."""
        if self.head_kind == "gmm":
            return [(table["fc%d_W" % l], table["fc%d_b" % l], "relu" if l < 4 else None) for l in (1, 2, 3, 4)]
        # one time step under 'same' padding meets only the centre tap of each (k, C_in, C_out) kernel (lstm.py:151)
        c = table["conv1_W"].shape[0] // 2
        return [(table["conv%d_W" % l][c], table["conv%d_b" % l], "relu" if l < 3 else "tanh") for l in (1, 2, 3)]

    def _head(self, h, head_masks=None):
        w = self.w
        if self.head_kind != "meanvar":      # -> [output of every layer]; the last one is the mixture (B,10n) / the second (B,3*fps)
            return ops.mlp_head_fwd(h, self._mlp_layers(w), masks=head_masks, n_mix=self.n_mix if self.head_kind == "gmm" else 0)
        if self._head_fused(h):      # one launch instead of seven (tf_head.hip)
            return ops.tf_head_fwd(h, w)
        a1 = ops.dense(h, w["mu_W1"], w["mu_b1"], activation=None)
        ops.act_fwd(a1, "relu", out=a1)
        mu = ops.dense(a1, w["mu_W2"], w["mu_b2"], activation="tanh")
        a3 = ops.dense(h, w["var_W1"], w["var_b1"], activation=None)
        ops.act_fwd(a3, "relu", out=a3)
        var = ops.dense(a3, w["var_W2"], w["var_b2"], activation=None)
        ops.act_fwd(var, "exp", out=var)
        return a1, mu, a3, var

    def predict(self, x, init_state=None):
        """(mu (B,3), var (B,3), state (L,2,B,H) as LSTMStateTuple (c,h) per layer) - inference, no dropout.
        head_kind 'gmm': (params (B,10n) = [pi | us | sigmas | rhos], None, state); 'raw': (second (B,3*fps), None, state)."""
        mu, var, states = self._predict_padded(x, self._pad_state(init_state))
        return mu, var, self._state_out(states)

    def _predict_padded(self, x, init_state=None):
        """init_state: (L,2,B,Hp) tensor, or a list of (c, h) pairs at width Hp, or None -> (mu, var, [(cT, hT)])."""
        inp, states = x, []
        if self._stack2(x):
            o = self._stack2_forward(x, init_state, reserve=False)
            states = [(o[0][2], o[0][1]), (o[1][2], o[1][1])]
        else:
            for l in range(self.L):
                c0 = None if init_state is None else init_state[l][0].contiguous()
                h0 = None if init_state is None else init_state[l][1].contiguous()
                hs, hT, cT = ops.lstm_seq(inp, self.w["K%d" % l], self.w["R%d" % l], self.w["b%d" % l], h0, c0, act="sigmoid",
                                          impl=self.impl, workspace=self.ws)
                states.append((cT, hT))
                inp = hs
        if self.head_kind != "meanvar":
            return self._head(states[-1][1])[-1], None, states
        _, mu, _, var = self._head(states[-1][1])
        return mu, var, states

    def rollout(self, x, init_state, noise):
        """Test-time loop of lstm.py:714-740: noise (P,B,3*fps) standard normal; each step predicts from the window and
        the state the previous step returned, then shifts in one second sampled around the prediction.
        -> (mus (P,B,3), vars (P,B,3), final state (L,2,B,H))."""
        P, B, F = noise.shape
        T = x.shape[1]
        mus = torch.empty((P, B, 3), dtype=torch.float32, device=self.device)
        vs = torch.empty((P, B, 3), dtype=torch.float32, device=self.device)
        win, st = x, self._pad_state(init_state)
        for k in range(P):
            mu, var, st = self._predict_padded(win, st)
            mus[k].copy_(mu); vs[k].copy_(var)
            nxt = torch.empty_like(win)
            nxt[:, :T - 1].copy_(win[:, 1:])
            ops.sample_refeed(mu, var, noise[k], out=nxt[:, T - 1], std="sqrt")
            win = nxt
        return mus, vs, self._state_out(st)

    def rollout_gmm(self, x, init_state, u, z):
        """GMM test loop of lstm.py:690-698,735-745,820-825: the window (the script feeds only the last second, T = 1) and
        the carried state predict a mixture, one second is drawn from it (ops.gmm3d_sample: u (P,B,fps) uniform, z
        (P,B,fps,3) normal) and shifted in.  -> (samples (P,B,3*fps), final state (L,2,B,H))."""
        assert self.head_kind == "gmm"
        P, B, fps = u.shape
        T = x.shape[1]
        outs = torch.empty((P, B, 3 * fps), dtype=torch.float32, device=self.device)
        win, st = x, self._pad_state(init_state)
        for k in range(P):
            params, _, st = self._predict_padded(win, st)
            nxt = torch.empty_like(win)
            if T > 1:
                nxt[:, :T - 1].copy_(win[:, 1:])
            ops.gmm3d_sample(params, u[k], z[k], out=nxt[:, T - 1])
            outs[k].copy_(nxt[:, T - 1])
            win = nxt
        return outs, self._state_out(st)

    def rollout_raw(self, x, init_state, P):
        """Raw-prediction test loop (lstm.py:747-757,820-825): the predicted second itself is shifted in.
        -> (predictions (P,B,3*fps), final state)."""
        assert self.head_kind == "raw"
        B, T, F = x.shape
        outs = torch.empty((P, B, F), dtype=torch.float32, device=self.device)
        win, st = x, self._pad_state(init_state)
        for k in range(P):
            pred, _, st = self._predict_padded(win, st)
            outs[k].copy_(pred)
            nxt = torch.empty_like(win)
            if T > 1:
                nxt[:, :T - 1].copy_(win[:, 1:])
            nxt[:, T - 1].copy_(pred)
            win = nxt
        return outs, self._state_out(st)

    def _stack2(self, x, masks=None):
        """Both layers as one launch (fov_lstm_stack2_fwd)?  Not with a dropout mask between the layers."""
        return (self.L == 2 and self.impl == "auto" and (masks is None or masks[0] is None)
                and ops.lstm_stack2_supported(x.shape[0], x.shape[1], x.shape[2], self.Hp))

    def _stack2_forward(self, x, init_state, reserve, out_state=None):
        w = self.w
        sts = [None if init_state is None else (init_state[l][1].contiguous(), init_state[l][0].contiguous()) for l in range(2)]
        fin = [None if out_state is None else (out_state[l, 1], out_state[l, 0]) for l in range(2)]      # (hT, cT) of LSTMStateTuple (c, h)
        return ops.lstm_stack2(x, (w["K0"], w["R0"], w["b0"]), (w["K1"], w["R1"], w["b1"]), sts[0], sts[1], act="sigmoid",
                               workspace=self.ws, reserve=reserve, final1=fin[0], final2=fin[1])

    def _stack_forward(self, x, init_state, masks, out_state=None):
        """out_state: a (L,2,B,Hp) buffer the layers' final (c, h) are written into directly (no stack / copy afterwards)."""
        w = self.w
        if self._stack2(x, masks):
            o1, o2 = self._stack2_forward(x, init_state, reserve=True, out_state=out_state)
            sts = [None if init_state is None else (init_state[l][1].contiguous(), init_state[l][0].contiguous()) for l in range(2)]
            tape = [(x, o1[0], o1[3], None if sts[0] is None else sts[0][0], None if sts[0] is None else sts[0][1]),
                    (o1[0], o2[0], o2[3], None if sts[1] is None else sts[1][0], None if sts[1] is None else sts[1][1])]
            return tape, [(o1[2], o1[1]), (o2[2], o2[1])]
        tape, inp, states = [], x, []
        for l in range(self.L):
            c0 = None if init_state is None else init_state[l, 0].contiguous()
            h0 = None if init_state is None else init_state[l, 1].contiguous()
            out = None
            if out_state is not None:
                e = lambda *shp: torch.empty(shp, dtype=torch.float32, device=x.device)
                out = (e(x.shape[0], x.shape[1], self.Hp), out_state[l, 1], out_state[l, 0], e(x.shape[0], x.shape[1], 5, self.Hp))
            hs, hT, cT, res = ops.lstm_seq_train(inp, w["K%d" % l], w["R%d" % l], w["b%d" % l], h0, c0, act="sigmoid",
                                                 impl=self.impl, workspace=self.ws, out=out)
            tape.append((inp, hs, res, h0, c0))
            states.append((cT, hT))
            inp = hs if (masks is None or l == self.L - 1) else hs * masks[l]
        return tape, states

    def _mlp_backward(self, hT, acts, dlast, accumulate, head_masks=None):
        gl = self._mlp_layers(self.g)
        return ops.mlp_head_bwd(hT, self._mlp_layers(self.w), acts, dlast, [g[0] for g in gl], [g[1] for g in gl], masks=head_masks,
                                need_dx=True, accumulate=accumulate, scratch=self.scratch)

    def _head_backward(self, hT, head, dmu, dvar, accumulate):
        w, g, sc = self.w, self.g, self.scratch
        if self._head_fused(hT):     # one launch instead of thirteen
            return ops.tf_head_bwd(hT, w, head, dmu, dvar, g, accumulate=accumulate)
        a1, mu, a3, var = head
        d2 = ops.act_bwd(dmu, mu, activation="tanh")
        da1, _, _ = ops.dense_bwd(a1, w["mu_W2"], d2, dW=g["mu_W2"], db=g["mu_b2"], scratch=sc, accumulate=accumulate)
        d1 = ops.act_bwd(da1, a1, activation="relu")
        dh_a, _, _ = ops.dense_bwd(hT, w["mu_W1"], d1, dW=g["mu_W1"], db=g["mu_b1"], scratch=sc, accumulate=accumulate)
        d4 = ops.act_bwd(dvar, var, activation="exp")
        da3, _, _ = ops.dense_bwd(a3, w["var_W2"], d4, dW=g["var_W2"], db=g["var_b2"], scratch=sc, accumulate=accumulate)
        d3 = ops.act_bwd(da3, a3, activation="relu")
        dh_b, _, _ = ops.dense_bwd(hT, w["var_W1"], d3, dW=g["var_W1"], db=g["var_b1"], scratch=sc, accumulate=accumulate)
        return ops.act_bwd(dh_b, hT, base=dh_a, activation=None)      # dh_a + dh_b

    def _stack2_bwd_ok(self, tape):
        """Asked at every step (a cheap C call): the answer depends on FOV_NO_STACK2 and the CU count the library sees, which
        ops._sync_env reloads live - a cached answer would send a later step into a launch that now refuses."""
        B, T, F = tape[0][0].shape
        return ops.lstm_stack2_bwd_supported(B, T, F, self.w["R0"].shape[0])

    def _stack_backward(self, tape, dhT, masks, accumulate, need_dx0=False):
        w, g = self.w, self.g
        dhs, dx0 = None, None
        # (Tried twice: each layer's weight-gradient products on a second stream beside the next layer's recurrence - at lstm.py's
        # batch a BPTT launch occupies 32 of 256 CUs.  Arranged from Python (shifted copy of hs + fov_wgrad_fused + two stream
        # hand-offs per layer) 0.596 -> 0.656 ms; arranged inside fov_lstm_seq_bwd (events, products deferred until the next
        # recurrence is queued) 0.520 -> 0.538 ms: the 45 us of products do overlap in the timeline, but every cross-stream
        # hand-off leaves a 7-16 us gap in a step that is a chain of short launches.)
        # Two layers, no dropout masks between them, no gradient towards the input: both recurrences and the data-gradient
        # product between them as ONE launch (fov_lstm_stack2_bwd: three roles on disjoint CUs; FOV_NO_STACK2=1: two launches)
        if self.L == 2 and masks is None and not need_dx0 and self._stack2_bwd_ok(tape):
            (x0, hs1, res1, h01, c01), (_, hs2, res2, h02, c02) = tape
            ops.lstm_stack2_bwd(x0, (w["K0"], w["R0"]), (w["K1"], w["R1"]), (hs1, res1, h01, c01), (hs2, res2, h02, c02), dhT2=dhT,
                                grads1=(g["K0"], g["R0"], g["b0"]), grads2=(g["K1"], g["R1"], g["b1"]), act="sigmoid",
                                accumulate=accumulate, scratch=self.bwd_scratch)
            return None
        for l in range(self.L - 1, -1, -1):
            inp, hs, res, h0, c0 = tape[l]
            b = ops.lstm_seq_bwd(inp, w["K%d" % l], w["R%d" % l], hs, res, h0=h0, c0=c0, dhs=dhs,
                                 dhT=dhT if l == self.L - 1 else None, dK=g["K%d" % l], dR=g["R%d" % l], db=g["b%d" % l],
                                 need_dx=(l > 0 or need_dx0), act="sigmoid", scratch=self.bwd_scratch, accumulate=accumulate)
            if l > 0:
                dhs = b["dx"] if masks is None else b["dx"] * masks[l - 1]
            else:
                dx0 = b["dx"]
        return dx0

    def _fb_gmm(self, x, y, init_state, masks, head_masks, grad_weight=1.0, state_view=False):
        """lstm.py:482-485: one window, costfunc.mixture_3d_gaussian_loss on y (second 0 under cfg.process_in_seconds,
        every frame of (B,T,3) otherwise), divided by batch_size * running_length [* fps] (cost.py:544-549)."""
        B = x.shape[0]
        pis = self.process_in_seconds
        n_pts = self.fps if pis else y.shape[1]
        # data parallelism: batch_size (cfg.batch_size) is the GLOBAL batch - the per-rank sums then add up to the script's loss
        # under the SUM all-reduce as they are; without it the divisor is this rank's B and the rank's share weighs it
        scale = (1.0 if self.batch_size else grad_weight) / ((self.batch_size or B) * self.running_length * (self.fps if pis else 1))
        init_state, masks = self._pad_state(init_state), self._pad_masks(masks)
        carry = self._carry_target(B, init_state) if state_view else None
        tape, states = self._stack_forward(x, init_state, masks, out_state=carry)
        hT = states[-1][1]
        acts = self._head(hT, head_masks)
        loss, dpre = ops.gmm3d_loss_grad(acts[-1], y, n_pts, scale, self.weight_by_pi, scratch=self.scratch)
        dhT = self._mlp_backward(hT, acts, dpre, accumulate=False, head_masks=head_masks)
        self._stack_backward(tape, dhT, masks, accumulate=False)
        return loss, acts[-1], None, (carry[..., :self.H] if state_view else self._state_out(states))

    def _fb_raw(self, x, y, init_state, masks, grad_weight=1.0):
        """lstm.py:486-508: prediction k is scored against second k (tf.losses.mean_squared_error over every element;
        pred_raw_loss_tf's total-variation term differences an axis of length one and is exactly zero, cost.py:608-618),
        then shifted into the window for prediction k+1 - no sampling, the gradient flows back through the predictions.
        use_reg adds 0.1 * sum (x^2+y^2+z^2-1)^2 (cost.py:622-641; the script itself leaves `use_reg` undefined)."""
        B, T, F = x.shape
        P = y.shape[1]
        sc = self.scratch
        init_state = self._pad_state(init_state)
        many = P > 1
        if many:
            self.grad.zero_()
        win, runs, total = x, [], None
        for k in range(P):
            if k > 0:
                nxt = torch.empty_like(win)
                if T > 1:
                    nxt[:, :T - 1].copy_(win[:, 1:])
                nxt[:, T - 1].copy_(runs[-1]["acts"][-1])
                win = nxt
            mk = None if masks is None else self._pad_masks(masks[k] if many else masks)
            tape, states = self._stack_forward(win, init_state, mk)
            hT = states[-1][1]
            acts = self._head(hT)
            pred = acts[-1]
            dP, loss = ops.mse_dense_grad(pred, y[:, k].contiguous(), activation=None, scratch=sc, weight=grad_weight)
            if self.use_reg:      # (a SUM over this rank's frames: ranks add up unweighted)
                dreg = torch.zeros_like(pred)
                reg = ops.xyz_sum1_grad(pred.view(B, F // 3, 3), dreg.view(B, F // 3, 3), scratch=sc)   # 0.5 * mean over the B*fps frames
                w = 0.2 * B * (F // 3)
                dP.add_(dreg, alpha=w)
                loss = loss + w * reg
            total = loss if total is None else ops.act_bwd(loss, loss, base=total, activation=None)
            runs.append({"tape": tape, "hT": hT, "acts": acts, "dP": dP, "masks": mk})
        for k in range(P - 1, -1, -1):
            r = runs[k]      # dP of prediction k is complete: its own loss plus every later window that holds it
            dpre = ops.act_bwd(r["dP"], r["acts"][-1], activation="tanh")
            dhT = self._mlp_backward(r["hT"], r["acts"], dpre, accumulate=many)
            dX = self._stack_backward(r["tape"], dhT, r["masks"], accumulate=many, need_dx0=(k > 0))
            for j in range(max(1, k - T + 1), k + 1):      # prediction j-1 sits in slot T-1-(k-j) of window k
                src = runs[j - 1]["dP"]
                slot = dX[:, T - 1 - (k - j)].contiguous()
                ops.act_bwd(slot, slot, base=src, activation=None, out=src)
        return total, runs[-1]["acts"][-1], None, self._state_out(states)

    def forward_backward(self, x, y, init_state=None, masks=None, noise=None, head_masks=None, grad_weight=1.0, state_view=False):
        """x (B,T,F), y (B,T_y,3*fps), init_state (L,2,B,H) (c,h) or None.  Fills self.grad (times grad_weight, a data-parallel
        rank's share of the global batch); returns
        (loss (1,), mu (B,3), var (B,3), final state (L,2,B,H)) - head_kind 'gmm': (loss, params (B,10n), None, state),
        'raw': (loss, last predicted second (B,3*fps), None, state).  head_masks: the two dropouts of _GMM_3dgassian
        [(B,64), (B,128)], pre-scaled; None = the script (tf.layers.dropout without training=True is the identity).

        noise None: the predict_len == 1 / is_test graph (lstm.py:430-434), one loss over all of y.
        noise (T_y-1, B, 3*fps) standard normal: the predict_len > 1 training graph (:446-468) - second k+1 is scored
        after re-running the whole stack, from the same fed state, on the window shifted by one second whose last slot
        is a second SAMPLED around the previous prediction (mean mu, stddev sqrt(var)); losses add up and the gradient
        flows back through the samples (reparameterisation, as TF differentiates tf.random_normal(mean, stddev)).
        `masks` is then a list of T_y per-window mask lists (DropoutWrapper draws a new mask per dynamic_rnn call).

        state_view (single-window graphs): the returned state is the width-H VIEW of one of the trainer's two carried-state
        buffers, which the kernels write directly (no gather / copy launch; fed back, it is taken as it is - no padding copy):
        the form lstm.py's loop wants (:612-620).  It is overwritten by the call after next; the default returns a copy."""
        if self.head_kind == "gmm":
            return self._fb_gmm(x, y, init_state, masks, head_masks, grad_weight, state_view)
        if self.head_kind == "raw":
            return self._fb_raw(x, y, init_state, masks, grad_weight)
        sc = self.scratch
        scale = grad_weight / (self.running_length * self.fps)
        init_state = self._pad_state(init_state)
        unpad = self._state_out
        if noise is None:
            masks = self._pad_masks(masks)
            carry = self._carry_target(x.shape[0], init_state) if state_view else None
            tape, states = self._stack_forward(x, init_state, masks, out_state=carry)
            hT = states[-1][1]
            head = self._head(hT)
            loss, dmu, dvar = ops.gauss_nll_grad(head[1], head[3], y, self.fps, scale, scratch=sc)
            dhT = self._head_backward(hT, head, dmu, dvar, accumulate=False)
            self._stack_backward(tape, dhT, masks, accumulate=False)
            return loss, head[1], head[3], (carry[..., :self.H] if state_view else unpad(states))
        B, T, F = x.shape
        P = y.shape[1]
        assert F == 3 * self.fps and tuple(noise.shape) == (P - 1, B, F)
        self.grad.zero_()
        win, runs, total = x, [], None
        for k in range(P):
            if k > 0:       # shift the window by one second, the new last second is a sample around prediction k-1
                nxt = torch.empty_like(win)
                nxt[:, :T - 1].copy_(win[:, 1:])
                ops.sample_refeed(runs[-1]["head"][1], runs[-1]["head"][3], noise[k - 1], out=nxt[:, T - 1], std="sqrt")
                win = nxt
            mk = None if masks is None else self._pad_masks(masks[k])
            tape, states = self._stack_forward(win, init_state, mk)
            hT = states[-1][1]
            head = self._head(hT)
            loss, dmu, dvar = ops.gauss_nll_grad(head[1], head[3], y[:, k:k + 1].contiguous(), self.fps, scale, scratch=sc)
            total = loss if total is None else ops.act_bwd(loss, loss, base=total, activation=None)
            runs.append({"tape": tape, "hT": hT, "head": head, "dmu": dmu, "dvar": dvar, "masks": mk})
        for k in range(P - 1, -1, -1):
            r = runs[k]     # dmu / dvar of prediction k are complete: loss k plus every later window holding sample k+1
            dhT = self._head_backward(r["hT"], r["head"], r["dmu"], r["dvar"], accumulate=True)
            dX = self._stack_backward(r["tape"], dhT, r["masks"], accumulate=True, need_dx0=(k > 0))
            for j in range(max(1, k - T + 1), k + 1):      # sample j (drawn from prediction j-1) sits in slot T-1-(k-j)
                src = runs[j - 1]
                ops.sample_refeed_bwd(dX[:, T - 1 - (k - j)], src["head"][3], noise[j - 1], src["dmu"], src["dvar"], std="sqrt")
        last = runs[-1]
        return total, last["head"][1], last["head"][3], unpad(states)

    def train_step(self, x, y, init_state=None, masks=None, noise=None, head_masks=None, n_global=None, state_view=False):
        """One step of lstm.py:612-620's sess.run([train_op, current_state]) -> (loss (1,), final state (L,2,B,H)) - the state is
        this rank's own (its sequences'), the loss the global one under data parallelism (n_global = global batch size).
        state_view: see forward_backward (the carried state without copies)."""
        weight = self._begin_step(x.shape[0], n_global)
        view = bool(state_view) and self.head_kind != "raw" and noise is None
        loss, _, _, state = self.forward_backward(x, y, init_state, masks, noise, head_masks, grad_weight=weight, state_view=view)
        return self._finish_step(loss), state

    def eval_loss(self, x, y, init_state=None, masks=None, noise=None, head_masks=None, state_view=False):
        """The `cost` fetch of the script's display step (lstm.py:625-650): loss and state of one batch, no update."""
        view = bool(state_view) and self.head_kind != "raw" and noise is None
        loss, _, _, state = self.forward_backward(x, y, init_state, masks, noise, head_masks, state_view=view)
        return loss, state

    # ---- tf.train.Saver's view of the model (lstm.py:552): every variable and its RMSProp slots under the graph's names ----
    def _tf_names(self, scope=""):
        """parameter key -> TF-1.x variable name of lstm.py's graph (MultiRNNCell under dynamic_rnn; contrib fully_connected /
        tf.layers.conv1d heads numbered in creation order)."""
        names = {}
        for l in range(self.L):
            names["W%d" % l] = "%srnn/multi_rnn_cell/cell_%d/lstm_cell/kernel" % (scope, l)
            names["B%d" % l] = "%srnn/multi_rnn_cell/cell_%d/lstm_cell/bias" % (scope, l)
        if self.head_kind == "raw":
            for i in range(3):
                sfx = "" if i == 0 else "_%d" % i
                names["conv%d_W" % (i + 1)] = "conv1d%s/kernel" % sfx
                names["conv%d_b" % (i + 1)] = "conv1d%s/bias" % sfx
        else:
            keys = (("mu_W1", "mu_b1"), ("mu_W2", "mu_b2"), ("var_W1", "var_b1"), ("var_W2", "var_b2")) if self.head_kind == "meanvar" \
                else tuple(("fc%d_W" % i, "fc%d_b" % i) for i in (1, 2, 3, 4))
            for i, (kw_, kb_) in enumerate(keys):
                sfx = "" if i == 0 else "_%d" % i
                names[kw_] = "fully_connected%s/weights" % sfx
                names[kb_] = "fully_connected%s/biases" % sfx
        return names

    def _cells_from(self, table, with_forget_bias):
        """[(W (F+H,4H), b (4H))] in tf.contrib LSTMCell layout from the flat table (self.w or the slot views)."""
        out = []
        for l in range(self.L):
            K, R, b = (self._unpadded(table, "%s%d" % (n, l)) for n in ("K", "R", "b"))
            H = R.shape[0]
            perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
            W = np.concatenate([K, R], 0)
            Wt, bt = np.empty_like(W), np.empty_like(b)
            Wt[:, perm] = W
            bk = b.copy()
            if with_forget_bias:
                bk[H:2 * H] -= np.float32(self.forget_bias)
            bt[perm] = bk
            out.append((Wt, bt))
        return out

    def state_dict(self, scope=""):
        """{TF variable name: array} of everything tf.train.Saver() writes for this graph: the LSTM cells (tf.contrib layout, the
        forget bias taken out again), the head, `<name>/RMSProp` (the rms slot, initialised to one) and `<name>/RMSProp_1` (the
        momentum slot: zeros, momentum is 0), the learning-rate variable and - not a TF variable - the update counter."""
        slot_views = {k: self.ms[self.offset[k]:self.offset[k] + self.w[k].numel()].view(self.w[k].shape) for k in self.order}
        names = self._tf_names(scope)
        out = {}
        for tag, table, fb in (("", self.w, True), ("/RMSProp", slot_views, False)):
            for l, (W, b) in enumerate(self._cells_from(table, fb)):
                out[names["W%d" % l] + tag], out[names["B%d" % l] + tag] = W, b
            for k in self.HEAD:
                out[names[k] + tag] = self._unpadded(table, k)
        for name in [n for n in out if not n.endswith("/RMSProp")]:
            out[name + "/RMSProp_1"] = np.zeros_like(out[name])
        out["Variable"] = np.float64(self.lr)          # lr = tf.Variable(cfg.LEARNING_RATE, trainable=False), lstm.py:554 (kept exact)
        out["fov/applied_updates"] = np.int64(self.step_count)
        return out

    def load_state_dict(self, sd, scope=""):
        """saver.restore (lstm.py:589-592,669-672): parameters, rms slots and learning rate back in place (padded slices stay
        exactly zero / one as a fresh trainer has them)."""
        from .models import convert_tf_lstmcell, pad_lstm
        names = self._tf_names(scope)

        def put(table_flat, k, arr):
            view = table_flat[self.offset[k]:self.offset[k] + self.w[k].numel()].view(self.w[k].shape)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            if tuple(arr.shape) != tuple(view.shape):      # zero-pad the H axis up to Hp (weights: 0; the rms slot of a padded
                pad = [(0, view.shape[i] - arr.shape[i]) for i in range(arr.ndim)]      # element is never read with a gradient)
                arr = np.pad(arr, pad, constant_values=0.0 if table_flat is self.flat else 1.0)
            view.copy_(torch.from_numpy(arr))

        for tag, flat, fb in (("", self.flat, self.forget_bias), ("/RMSProp", self.ms, 0.0)):
            for l in range(self.L):
                W, b = sd[names["W%d" % l] + tag], sd[names["B%d" % l] + tag]
                K, R, bk = convert_tf_lstmcell(np.asarray(W, np.float32), np.asarray(b, np.float32), fb)
                if self.Hp != self.H:
                    Kp, Rp, bp = pad_lstm(K, R, bk, self.Hp, pad_input=(l > 0))
                    if tag:       # slot of a padded element: one, as TF initialises it (its gradient is exactly zero: never used)
                        ones = pad_lstm(np.ones_like(K), np.ones_like(R), np.ones_like(bk), self.Hp, pad_input=(l > 0))
                        Kp, Rp, bp = (np.where(o == 1.0, v, np.float32(1.0)) for v, o in zip((Kp, Rp, bp), ones))
                    K, R, bk = Kp, Rp, bp
                put(flat, "K%d" % l, K); put(flat, "R%d" % l, R); put(flat, "b%d" % l, bk)
            for k in self.HEAD:
                put(flat, k, sd[names[k] + tag])
        self.lr = float(sd["Variable"])
        self.step_count = int(sd.get("fov/applied_updates", 0))
        self.applied.fill_(self.step_count)
