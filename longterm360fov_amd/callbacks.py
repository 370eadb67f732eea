"""Keras-style callbacks used by the reference's training scripts (mycode/FoV_seq2seq.py:108-111,
given_others_gt_mean_var_seq2seq.py:484-487): ModelCheckpoint, ReduceLROnPlateau, EarlyStopping.
They observe the `logs` dict of each epoch ({'loss', 'val_loss', 'lr'}) and act on the model."""
import numpy as np


class Callback:
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


def _monitor_op(mode, monitor):
    if mode == "max" or (mode == "auto" and "acc" in monitor):
        return np.greater, -np.inf
    return np.less, np.inf


class ModelCheckpoint(Callback):
    """Saves weights to `filepath.format(epoch=epoch+1, **logs)`: '.h5' names as Keras-layout HDF5 (keras_h5.py),
    anything else as '.npz' (same arrays, same order)."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, mode="auto", period=1):
        self.filepath, self.monitor, self.save_best_only, self.period = filepath, monitor, save_best_only, period
        self.op, self.best = _monitor_op(mode, monitor)
        self.saved = []
        self._since = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self._since += 1
        if self._since < self.period:
            return
        self._since = 0
        path = self.model.weights_path(self.filepath.format(epoch=epoch + 1, **logs))
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.op(cur, self.best):
                return
            self.best = cur
        self.model.save_weights(path)
        self.saved.append(path)


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto", min_delta=1e-4,
                 cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        self.monitor, self.factor, self.patience, self.min_delta = monitor, factor, patience, min_delta
        self.cooldown, self.min_lr, self.mode = cooldown, min_lr, mode
        self._reset()

    def _reset(self):
        if self.mode == "max" or (self.mode == "auto" and "acc" in self.monitor):
            self.op, self.best = (lambda a, b: np.greater(a, b + self.min_delta)), -np.inf
        else:
            self.op, self.best = (lambda a, b: np.less(a, b - self.min_delta)), np.inf
        self.cooldown_counter = 0
        self.wait = 0

    def on_train_begin(self, logs=None):
        self._reset()

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs["lr"] = self.model.lr
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.op(cur, self.best):
            self.best = cur
            self.wait = 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = float(self.model.lr)
                if old > self.min_lr:
                    self.model.lr = max(old * self.factor, self.min_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto"):
        self.monitor, self.patience, self.min_delta = monitor, patience, abs(min_delta)
        self.op, _ = _monitor_op(mode, monitor)
        if self.op is np.greater:
            self.min_delta *= 1
        else:
            self.min_delta *= -1
        self.stopped_epoch = 0

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        self.best = np.inf if self.op is np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.op(cur - self.min_delta, self.best):
            self.best = cur
            self.wait = 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)
