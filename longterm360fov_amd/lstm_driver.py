"""The driver of mycode/lstm.py around training.TFLSTMTrainer: the epoch loop with the LSTM state carried across
batches and the learning-rate halving schedule (lstm.py:583-660), tf.train.Saver-style save / restore of every variable AND
the optimizer's slots (:552-553,589-598,659,669-672), and the test loop that rolls the model forward from one second of
history (:663-828).  Data comes from the reference's own data layer, which stays as it is (dataLayer2.DataLayer:
`_get_next_minibatch(datadb, batch_size)` -> (batch_x (B,T,90), batch_y, others_future, batch_y_further, db_index,
others_future_further)) or from any callable returning such tuples; all arithmetic runs in libfov360_hip.so through the
trainer.

Checkpoints: one '.npz' per save holding the arrays tf.train.Saver would write for this graph under TF's variable names
(`rnn/multi_rnn_cell/cell_0/lstm_cell/kernel`, `.../kernel/RMSProp`, `fully_connected/weights`, `Variable` (the learning
rate), ...; TFLSTMTrainer.state_dict) - TensorFlow itself is not available to write its own container format.  Under data
parallelism the replicas are bit-identical: rank 0 writes through a temporary file, every rank restores from the same file."""
import os

import numpy as np
import torch

from . import parallel


def total_batch_of(datadb, cfg, batch_size=None, fps=None):
    """Steps per epoch as lstm.py:570-580 counts them: frames of all training videos (cfg.test_video_ind excluded) minus one
    window, over the chunk stride and the batch size, times the users of the last video.  (The script leaves the value a float
    under Python 3, where `range(total_batch)` raises; the integer part is what Python 2 computed.)"""
    fps = fps or cfg.get("fps", 30)
    batch_size = batch_size or cfg.get("batch_size", 32)
    frames, users = 0, 0
    for key in datadb.keys():
        users = datadb[key]["x"].shape[0]
        if key == cfg.get("test_video_ind", None):
            continue
        frames += datadb[key]["x"].shape[1]
    window = cfg.get("running_length", 10) * (fps if cfg.get("process_in_seconds", True) else 1)
    return int((frames - window) / cfg.get("data_chunk_stride", 10) / batch_size) * users


class LSTMPyDriver:
    """trainer: a training.TFLSTMTrainer (any head).  model_path: the script's "./model/LSTM_<tag>.ckpt"; epoch checkpoints
    are "<model_path minus '.ckpt'>epoch<N>.ckpt.npz" as the script names them (:597,659).  dropout: the value the script feeds
    the DropoutWrapper placeholder while training (0.1, :616) - masks are drawn per step from a seeded device generator
    (seed + rank); 0 = no masks (the two layers then run as one launch)."""

    def __init__(self, trainer, cfg=None, model_path="./model/LSTM_fov.ckpt", dropout=0.1, seed=0, log=None):
        from .config import cfg as default_cfg
        self.trainer, self.cfg = trainer, (cfg if cfg is not None else default_cfg)
        self.model_path, self.dropout = model_path, float(dropout)
        self.base_lr = float(self.cfg.get("LEARNING_RATE", trainer.lr))
        self.lr_epoch_step = float(self.cfg.get("lr_epoch_step", 10))
        self.log = log if log is not None else (lambda msg: None)
        self.history = []          # (count, loss) of every display step, what the script prints and writes as a summary
        self.saved = []
        self._gen = torch.Generator(device=trainer.device)
        self._gen.manual_seed(int(seed) + parallel.world()[0])

    # ---- tf.train.Saver -------------------------------------------------------------------------------------------------
    def epoch_path(self, epoch):
        return self.model_path[:-5] + "epoch" + str(epoch) + ".ckpt"

    @staticmethod
    def _file(path):
        return path if path.endswith(".npz") else path + ".npz"

    def save(self, path):
        """saver.save(sess, path): variables, RMSProp slots, learning rate.  Rank 0 writes (temporary file, then rename)."""
        path = self._file(path)
        if parallel.world()[0] == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = "%s.tmp%d" % (path, os.getpid())
            with open(tmp, "wb") as f:
                np.savez(f, **self.trainer.state_dict())
            os.replace(tmp, path)
        if parallel.dp_active():
            torch.distributed.barrier()      # nobody reads the file (restore on every rank) before it is complete
        self.saved.append(path)
        return path

    def restore(self, path):
        """saver.restore(sess, path) on every rank."""
        with np.load(self._file(path)) as z:
            self.trainer.load_state_dict({k: z[k] for k in z.files})

    def exists(self, path):
        return os.path.exists(self._file(path))      # (the script globs for TF's '.meta' file, :590)

    # ---- training loop ---------------------------------------------------------------------------------------------------
    def _masks(self, B, T):
        if self.dropout <= 0.0:
            return None
        keep = 1.0 - self.dropout
        tr = self.trainer
        return [(torch.rand((B, T, tr.H), generator=self._gen, device=tr.device) < keep).to(torch.float32) / keep
                for _ in range(tr.L - 1)]

    @staticmethod
    def _next(source, datadb, batch_size):
        if hasattr(source, "_get_next_minibatch"):
            return source._get_next_minibatch(datadb, batch_size)
        return source()

    def _dev(self, a):
        if torch.is_tensor(a):
            return a.to(device=self.trainer.device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.trainer.device)

    def fit(self, data_io, total_batch, training_epochs=5, batch_size=None, datadb=None, starting_epoch=2, n_global=None,
            check_every_epoch=True):
        """lstm.py:583-660.  The quirks are the script's: epochs run from `starting_epoch` = 2 (or from `training_epochs` when
        the checkpoint of epoch starting_epoch - 1 exists and is restored, :590-592); on every even epoch the model is saved
        and the rate set to LEARNING_RATE * 0.5 ** (epoch / lr_epoch_step) BEFORE the epoch's steps; the state the last step
        returned is the next step's initial state (zeros at the very start only); a display step re-evaluates the batch and, on
        the mean / variance branch, its state replaces the carried one (:625-637).  After the last epoch the model is saved
        once more.  -> the carried state (L,2,B,H)."""
        tr = self.trainer
        batch_size = batch_size or self.cfg.get("batch_size", 32)
        if self.exists(self.epoch_path(starting_epoch - 1)):
            self.restore(self.epoch_path(starting_epoch - 1))
            self.log("Model restored.")
            starting_epoch = training_epochs
        state = torch.zeros((tr.L, 2, batch_size, tr.H), dtype=torch.float32, device=tr.device)
        epoch = starting_epoch
        for epoch in range(starting_epoch, starting_epoch + training_epochs):
            if epoch > 0 and epoch % 2 == 0:
                self.log("Model saved: %s" % self.save(self.epoch_path(epoch)))
                tr.lr = self.base_lr * (0.5 ** (epoch / self.lr_epoch_step))
                self.log("epoch: %d , change lr=lr*0.5, lr= %g" % (epoch, tr.lr))
            for step in range(total_batch):
                batch = self._next(data_io, datadb, batch_size)
                x, y = self._dev(batch[0]), self._dev(batch[1])
                masks = self._masks(x.shape[0], x.shape[1])
                _, state = tr.train_step(x, y, state, masks=masks, n_global=n_global, state_view=True)      # (the carried state without copies)
                count = (step + 1) * batch_size + epoch * total_batch * batch_size
                display_step = 10 if count < 200 else 200
                if count % display_step == 0:
                    loss, st = tr.eval_loss(x, y, state, masks=self._masks(x.shape[0], x.shape[1]), state_view=True)
                    if tr.head_kind == "meanvar":
                        state = st
                    self.history.append((count, float(loss.item())))
                    self.log("Step %d, Minibatch Loss= %.6f" % self.history[-1])
            if check_every_epoch:
                tr.check()       # fail-stop: a persistent kernel that gave up (on any rank) ends the run here, updates were skipped
        self.log("Optimization Finished!")
        self.log("Model saved: %s" % self.save(self.epoch_path(epoch)))
        return state

    # ---- test loop -------------------------------------------------------------------------------------------------------
    def test(self, data_io, num_trials, predict_step=None, batch_size=None, datadb=None, test_epoch=1, seed=1234):
        """lstm.py:663-828: restore the checkpoint of `test_epoch` (None: model_path itself) if it exists, then per trial start from
        a ZERO state, feed ONLY the last second of the window (:687) and roll forward predict_step seconds - sampling around the
        predicted mean / from the predicted mixture / re-feeding the raw prediction, each inside the trainer's rollout.
        -> (test_out: list of (predict_step, B, ...) arrays per trial, gt_out: list of batch_y_further per trial)."""
        tr = self.trainer
        predict_step = predict_step or self.cfg.get("predict_step", 10)
        batch_size = batch_size or self.cfg.get("batch_size", 32)
        path = self.model_path if test_epoch is None else self.epoch_path(test_epoch)
        if self.exists(path):
            self.restore(path)
            self.log("Model restored.")
        gen = torch.Generator(device=tr.device)
        gen.manual_seed(int(seed))
        test_out, gt_out = [], []
        for _ in range(num_trials):
            batch = self._next(data_io, datadb, batch_size)
            x = self._dev(batch[0])[:, -1:, :].contiguous()
            gt_out.append(np.asarray(batch[3]))
            B, F = x.shape[0], x.shape[2]
            fps = F // 3
            state = torch.zeros((tr.L, 2, B, tr.H), dtype=torch.float32, device=tr.device)
            if tr.head_kind == "meanvar":
                noise = torch.randn((predict_step, B, F), generator=gen, device=tr.device)
                mus, vs, _ = tr.rollout(x, state, noise)
                out = torch.cat([mus, vs], dim=2)                      # (P,B,6): [ux uy uz varx vary varz] as :727-733 stacks them
            elif tr.head_kind == "gmm":
                u = torch.rand((predict_step, B, fps), generator=gen, device=tr.device)
                z = torch.randn((predict_step, B, fps, 3), generator=gen, device=tr.device)
                out, _ = tr.rollout_gmm(x, state, u, z)
            else:
                out, _ = tr.rollout_raw(x, state, predict_step)
            test_out.append(out.cpu().numpy())
        tr.check()
        return test_out, gt_out
