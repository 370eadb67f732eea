"""One training step of the teacher-forced seq2seq graph on the GPU (model.fit's inner loop,
mycode/FoV_seq2seq.py:103,112-117): forward with reserve -> MSE -> Dense backward -> decoder BPTT
-> encoder BPTT -> (data-parallel: ONE all-reduce of the flat gradient buffer) -> Keras Adam /
RMSprop on the flat parameter buffer.  All arithmetic happens in libfov360_hip.so; torch holds
the buffers and runs the RCCL collective."""
import numpy as np
import torch

from . import ops, parallel

_W_ORDER = ("enc_K", "enc_R", "enc_b", "dec_K", "dec_R", "dec_b", "dense_W", "dense_b")


class Seq2SeqTrainer:
    def __init__(self, weights, act="sigmoid", impl="auto", optimizer="adam", lr=1e-3, device="cuda"):
        self.act, self.impl, self.optimizer, self.lr, self.device = act, impl, optimizer, float(lr), device
        self.shapes = [(k, tuple(weights[k].shape)) for k in _W_ORDER]
        n = int(sum(np.prod(s) for _, s in self.shapes))
        self.flat = torch.empty(n, dtype=torch.float32, device=device)     # parameters, one buffer
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)     # gradients, same layout
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device) if optimizer == "adam" else None
        self.w, self.g = {}, {}
        off = 0
        for k, s in self.shapes:
            cnt = int(np.prod(s))
            self.w[k] = self.flat[off:off + cnt].view(*s)
            self.g[k] = self.grad[off:off + cnt].view(*s)
            self.w[k].copy_(torch.from_numpy(np.ascontiguousarray(weights[k], dtype=np.float32)))
            off += cnt
        self.step_count = 0
        self.ws = ops.Workspace()
        self.scratch = ops.Scratch()
        self._bufs = {}

    def weights_numpy(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.w.items()}

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
                "dpre": e(B, T_out, O), "loss": torch.zeros(1, dtype=torch.float32, device=self.device),
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
        y = ops.dense(dhs_, w["dense_W"], w["dense_b"], activation="tanh")
        dpre, loss = ops.mse_dense_grad(y, target, "tanh", scratch=self.scratch, dpre=bufs["dpre"], loss=bufs["loss"])
        d_hs, _, _ = ops.dense_bwd(dhs_, w["dense_W"], dpre, dW=g["dense_W"], db=g["dense_b"], scratch=self.scratch)
        bd = ops.lstm_seq_bwd(dec_in, w["dec_K"], w["dec_R"], dhs_, dres, h0=ehT, c0=ecT, dhs=d_hs, dK=g["dec_K"],
                              dR=g["dec_R"], db=g["dec_b"], need_state_grads=True, act=self.act, dz=bufs["dz_dec"],
                              scratch=self.scratch)
        ops.lstm_seq_bwd(enc, w["enc_K"], w["enc_R"], ehs, eres, dhT=bd["dh0"], dcT=bd["dc0"], dK=g["enc_K"],
                         dR=g["enc_R"], db=g["enc_b"], act=self.act, dz=bufs["dz_enc"], scratch=self.scratch)
        if grad_weight != 1.0:
            self.grad.mul_(grad_weight)
        return loss, y

    def apply_gradients(self):
        self.step_count += 1
        if self.optimizer == "adam":
            ops.adam_step(self.flat, self.grad, self.m, self.v, self.step_count, lr=self.lr)
        else:
            ops.rmsprop_step(self.flat, self.grad, self.m, lr=self.lr)

    def train_step(self, enc, dec_in, target, n_global=None):
        """One optimizer step.  Under data parallelism every rank passes its shard of the global
        batch and `n_global` = global batch size: gradients are combined as sum_r (n_r/n) g_r with a
        single all-reduce of the flat buffer, so the update equals the single-process one."""
        _, world = parallel.world()
        n_local = enc.shape[0]
        weight = 1.0 if world == 1 else n_local / float(n_global if n_global else n_local * world)
        loss, _ = self.forward_backward(enc, dec_in, target, grad_weight=weight)
        if world > 1:
            torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM)
            lw = loss * weight
            torch.distributed.all_reduce(lw, op=torch.distributed.ReduceOp.SUM)
            loss = lw
        self.apply_gradients()
        return loss

    def eval_loss(self, enc, dec_in, target):
        y = ops.seq2seq_teacher_forced(enc, dec_in, self.w, act=self.act, impl=self.impl, workspace=self.ws)
        _, loss = ops.mse_dense_grad(y, target, None, scratch=self.scratch)
        return loss
