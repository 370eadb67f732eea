"""mycode/lstm.py's driver around the training step (lstm.py:552-660, 663-828): TFLSTMTrainer as a FlatParamTrainer (fail-stop
check(), guarded optimizer, data-parallel all-reduce equal to the single-process step), lstm_driver.LSTMPyDriver's epoch loop
(state carried across batches, learning-rate schedule, save on even epochs), tf.train.Saver-style save / restore of weights AND
RMSProp slots, and the test loop.  The epoch loop is checked against a transcription of the script's loop written here, step
by step, bit for bit; the steps themselves are checked against torch.autograd in test_gpu_train.py / test_gpu_lstm_py_heads.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F, FPS = 90, 30


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _model(seed, H, kind):
    from test_gpu_lstm_py_heads import _cells, _gmm_head, _raw_head
    rng = np.random.default_rng(seed)
    cells = _cells(rng, F, H)
    if kind == "gmm":
        head = _gmm_head(rng, H)
    elif kind == "raw":
        head = _raw_head(rng, H)
    else:
        head = {}
        for br in ("mu", "var"):
            head[br + "_W1"] = (rng.standard_normal((H, 32)) / np.sqrt(H)).astype(np.float32)
            head[br + "_b1"] = (0.1 * rng.standard_normal(32)).astype(np.float32)
            head[br + "_W2"] = (rng.standard_normal((32, 3)) / np.sqrt(32)).astype(np.float32)
            head[br + "_b2"] = (0.1 * rng.standard_normal(3)).astype(np.float32)
    return cells, head


def _trainer(seed, H, kind, **kw):
    from longterm360fov_amd.training import TFLSTMTrainer
    cells, head = _model(seed, H, kind)
    kw.setdefault("lr", 1e-3)
    return TFLSTMTrainer(cells, head, head_kind=kind, fps=FPS, running_length=10, **kw)


def _batch(seed, B, T, Ty=1):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    y = rng.uniform(-1, 1, (B, Ty, F)).astype(np.float32)
    init = (0.2 * rng.standard_normal((2, 2, B, 400))).astype(np.float32)
    return x, y, init


class FakeDataLayer:
    """Stands in for dataLayer2.DataLayer (which stays the reference's): deterministic minibatches through the same method."""

    def __init__(self, seed, T=10, Ty=1, fixed=False):
        self.rng, self.T, self.Ty, self.calls, self.fixed, self.seed = np.random.default_rng(seed), T, Ty, 0, fixed, seed

    def _get_next_minibatch(self, datadb, batch_size):
        self.calls += 1
        if self.fixed:       # the same minibatch every time
            self.rng = np.random.default_rng(self.seed)
        x = self.rng.uniform(-1, 1, (batch_size, self.T, F)).astype(np.float32)
        y = self.rng.uniform(-1, 1, (batch_size, self.Ty, F)).astype(np.float32)
        further = self.rng.uniform(-1, 1, (batch_size, 10, F)).astype(np.float32)
        return x, y, None, further, None, None


def _poison(ws):
    ws.buf[:4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("kind", ["meanvar", "gmm", "raw"])
def test_tf_lstm_trainer_is_fail_stop(kind):
    """lstm.py's training step on the FlatParamTrainer base: two good steps, a give-up, two steps skipped ON THE DEVICE (parameters
    and the RMSProp slot untouched), check() raises once and counts back to the updates that ran; the following step equals an
    undisturbed third step bit for bit."""
    from longterm360fov_amd import _lib
    from longterm360fov_amd.training import FlatParamTrainer
    x, y, init = _batch(5, 12, 3)
    a = (dev(x), dev(y), dev(init))
    ref = _trainer(7, 400, kind)
    assert isinstance(ref, FlatParamTrainer)
    for _ in range(3):
        ref.train_step(*a)
    ref.check()
    tr = _trainer(7, 400, kind)
    tr.train_step(*a); tr.train_step(*a)
    after2, ms2 = tr.flat.clone(), tr.ms.clone()
    _poison(tr.ws)
    tr.train_step(*a); tr.train_step(*a)
    assert tr.step_count == 4 and torch.equal(tr.flat, after2) and torch.equal(tr.ms, ms2)
    with pytest.raises(_lib.FovError) as ei:
        tr.check()
    assert ei.value.code == _lib.ERR_TIMEOUT and tr.step_count == 2 and int(tr.applied.item()) == 2
    tr.train_step(*a)
    tr.check()
    assert tr.step_count == 3 and torch.equal(tr.flat, ref.flat) and torch.equal(tr.ms, ref.ms)


def _dp_worker(rank, world_size, port, q, kind, poison_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from longterm360fov_amd import _lib, parallel
        B = 21
        x, y, init = _batch(15, B, 3, Ty=(2 if kind == "raw" else 1))
        lo, hi = parallel.shard_range(B)
        a = (dev(x[lo:hi]), dev(y[lo:hi]), dev(init[:, :, lo:hi]))
        tr = _trainer(17, 400, kind, batch_size=(B if kind == "gmm" else None))
        losses, state = [], None
        for _ in range(3):
            loss, state = tr.train_step(*a, n_global=B)
            losses.append(float(loss.item()))
        tr.check()
        raised = None
        if poison_rank is not None:       # a give-up in a forward that NO all-reduce follows (a validation pass): every rank must raise
            if rank == poison_rank:
                _poison(tr.ws)
            try:
                tr.check()
            except _lib.FovError as e:
                raised = e.code
            tr.train_step(*a, n_global=B)     # the group is still in step
            tr.check()
        q.put((rank, losses, tr.flat.detach().cpu().numpy(), tr.ms.detach().cpu().numpy(), state.cpu().numpy(), raised))
    finally:
        dist.destroy_process_group()


def _run_two_ranks(kind, poison_rank=None):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, kind, poison_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("kind", ["meanvar", "gmm", "raw"])
def test_tf_lstm_trainer_two_ranks_equal_single_process(kind):
    """lstm.py's step under data parallelism (2 gloo ranks on the box's one GPU, shards 11 / 10): ONE SUM all-reduce of
    [poison | gradients | loss], the clip and the RMSProp update on the all-reduced gradient - three steps give the losses,
    parameters and rms slots of the single-process run on the whole batch; each rank's carried state is its shard's; the
    replicas stay bit-identical."""
    res = _run_two_ranks(kind)
    B = 21
    x, y, init = _batch(15, B, 3, Ty=(2 if kind == "raw" else 1))
    tr = _trainer(17, 400, kind, batch_size=(B if kind == "gmm" else None))
    ref_losses, state = [], None
    for _ in range(3):
        loss, state = tr.train_step(dev(x), dev(y), dev(init))
        ref_losses.append(float(loss.item()))
    ref_flat, ref_ms, ref_state = tr.flat.cpu().numpy(), tr.ms.cpu().numpy(), state.cpu().numpy()
    bounds = [(0, 11), (11, 21)]
    for (rank, losses, flat, ms, st, _), (lo, hi) in zip(res, bounds):
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
        d = np.abs(flat - ref_flat)
        assert d.max() <= 3e-3 and np.mean(d <= 2e-5) >= 0.995, (rank, d.max(), np.mean(d <= 2e-5))
        assert np.abs(ms - ref_ms).max() <= 1e-4
        assert np.abs(st - ref_state[:, :, lo:hi]).max() <= 2e-3
    np.testing.assert_array_equal(res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][3], res[1][3])


def test_give_up_outside_a_training_step_raises_on_every_rank():
    """Round-4 advisor finding: a give-up during the forwards between the last step and check() poisoned only the local
    workspace - that rank raised, its peers walked into the next all-reduce.  check() now all-reduces a one-element flag: rank 1
    poisons its workspace after the last step, BOTH ranks raise ERR_TIMEOUT, and the next step (a collective) completes."""
    from longterm360fov_amd import _lib
    res = _run_two_ranks("meanvar", poison_rank=1)
    for rank, _, _, _, _, raised in res:
        assert raised == _lib.ERR_TIMEOUT, (rank, raised)
    np.testing.assert_array_equal(res[0][2], res[1][2])


def _script_loop(tr, data_io, total_batch, training_epochs, batch_size, base_lr, lr_epoch_step, starting_epoch=2):
    """lstm.py:583-660 transcribed with the trainer's calls in place of sess.run - the specification the driver is held to."""
    saves, history = [], []
    state = torch.zeros((2, 2, batch_size, tr.H), device="cuda")
    for epoch in range(starting_epoch, starting_epoch + training_epochs, 1):
        if epoch > 0 and epoch % 2 == 0:
            saves.append((epoch, tr.flat.clone(), tr.ms.clone(), tr.lr))
            tr.lr = base_lr * (0.5 ** (epoch / lr_epoch_step))
        for step in range(total_batch):
            bx, by = data_io._get_next_minibatch(None, batch_size)[:2]
            _, state = tr.train_step(dev(bx), dev(by), state)
            count = (step + 1) * batch_size + epoch * total_batch * batch_size
            display_step = 10 if count < 200 else 200
            if count % display_step == 0:
                loss, st = tr.eval_loss(dev(bx), dev(by), state)
                if tr.head_kind == "meanvar":
                    state = st
                history.append((count, float(loss.item())))
    saves.append((epoch, tr.flat.clone(), tr.ms.clone(), tr.lr))
    return state, saves, history


@pytest.mark.parametrize("kind", ["meanvar", "gmm"])
def test_epoch_loop_equals_the_scripts_loop_and_checkpoints_round_trip(kind, tmp_path):
    """LSTMPyDriver.fit against the transcription above (no dropout so that both are deterministic): final parameters, rms
    slots, carried state, display-step losses, learning rate and the checkpoint of every even epoch, bit for bit; then
    saver.restore into a FRESH trainer gives back parameters, slots and rate exactly, and training on from the restored trainer
    equals training on from the original."""
    from longterm360fov_amd.config import default_config
    from longterm360fov_amd.lstm_driver import LSTMPyDriver
    cfg = default_config()
    cfg.LEARNING_RATE, cfg.lr_epoch_step, cfg.batch_size = 2e-3, 10, 10
    B, total_batch, epochs = 10, 5, 3
    ref = _trainer(31, 400, kind, lr=cfg.LEARNING_RATE, batch_size=B)
    ref_state, ref_saves, ref_hist = _script_loop(ref, FakeDataLayer(41), total_batch, epochs, B, cfg.LEARNING_RATE, cfg.lr_epoch_step)
    tr = _trainer(31, 400, kind, lr=cfg.LEARNING_RATE, batch_size=B)
    drv = LSTMPyDriver(tr, cfg, model_path=str(tmp_path / "LSTM_t.ckpt"), dropout=0.0)
    state = drv.fit(FakeDataLayer(41), total_batch, training_epochs=epochs, batch_size=B)
    assert torch.equal(tr.flat, ref.flat) and torch.equal(tr.ms, ref.ms) and torch.equal(state, ref_state)
    assert drv.history == ref_hist and len(ref_hist) >= 3
    assert tr.lr == ref.lr == cfg.LEARNING_RATE * 0.5 ** (4 / 10)            # epochs 2, 3, 4: halving schedule applied at 2 and 4
    assert [os.path.basename(p) for p in drv.saved] == ["LSTM_tepoch2.ckpt.npz", "LSTM_tepoch4.ckpt.npz", "LSTM_tepoch4.ckpt.npz"]
    # the checkpoint written at the START of epoch 4 was overwritten by the final one (same name, as in the script); epoch 2's is
    # the model before any step at the initial rate
    fresh = _trainer(99, 400, kind, lr=123.0, batch_size=B)
    d2 = LSTMPyDriver(fresh, cfg, model_path=str(tmp_path / "LSTM_t.ckpt"), dropout=0.0)
    d2.restore(d2.epoch_path(2))
    e2 = ref_saves[0]
    assert torch.equal(fresh.flat, e2[1]) and torch.equal(fresh.ms, e2[2]) and fresh.lr == e2[3] and fresh.padded_slices_are_zero()   # (saved before any step: every slot is still one)
    d2.restore(d2.epoch_path(4))
    assert torch.equal(fresh.flat, tr.flat) and fresh.lr == tr.lr and fresh.step_count == tr.step_count
    sa, sb = tr.state_dict(), fresh.state_dict()      # (the rms slot of a zero-PADDED element just decays in `tr` and restarts at one in
    assert sorted(sa) == sorted(sb)                   # `fresh`: its gradient is exactly zero, the value is never used - compare at width H)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # the file holds what tf.train.Saver would: TF names, tf.contrib layout, slots
    with np.load(d2.epoch_path(4) + ".npz") as z:
        names = set(z.files)
        W = z["rnn/multi_rnn_cell/cell_0/lstm_cell/kernel"]
        assert W.shape == (F + 400, 1600) and z["rnn/multi_rnn_cell/cell_1/lstm_cell/bias/RMSProp"].shape == (1600,)
        assert "fully_connected/weights" in names and "fully_connected_3/biases/RMSProp_1" in names and "Variable" in names
        for (W1, b1), l in zip(tr.cells_tf(), range(2)):
            np.testing.assert_array_equal(z["rnn/multi_rnn_cell/cell_%d/lstm_cell/kernel" % l], W1)
            np.testing.assert_array_equal(z["rnn/multi_rnn_cell/cell_%d/lstm_cell/bias" % l], b1)
    x, y, init = _batch(77, B, 10)
    for _ in range(2):
        la, sa = tr.train_step(dev(x), dev(y), dev(init))
        lb, sb = fresh.train_step(dev(x), dev(y), dev(init))
        assert torch.equal(la, lb) and torch.equal(sa, sb)
    assert torch.equal(fresh.flat, tr.flat)
    sa, sb = tr.state_dict(), fresh.state_dict()
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # lstm.py:590-592: the checkpoint of epoch starting_epoch - 1 exists -> restored, and the loop starts at `training_epochs`
    os.replace(d2.epoch_path(2) + ".npz", d2.epoch_path(1) + ".npz")
    again = _trainer(5, 400, kind, lr=cfg.LEARNING_RATE, batch_size=B)
    d3 = LSTMPyDriver(again, cfg, model_path=str(tmp_path / "LSTM_t.ckpt"), dropout=0.0)
    d3.fit(FakeDataLayer(43), 1, training_epochs=1, batch_size=B)
    assert [os.path.basename(p) for p in d3.saved] == ["LSTM_tepoch1.ckpt.npz"]      # epoch 1 is odd: only the final save


def test_dropout_masks_training_and_total_batch():
    """The driver at the script's settings (dropout 0.1 -> DropoutWrapper masks on what layer 1 hands up, drawn per step from
    a seeded generator): two runs with the same seed agree bit for bit, another seed differs; and
    total_batch_of counts lstm.py:570-580's steps."""
    from longterm360fov_amd.config import default_config
    from longterm360fov_amd.lstm_driver import LSTMPyDriver, total_batch_of
    cfg = default_config()
    cfg.LEARNING_RATE = 2e-3
    outs = []
    for seed in (3, 3, 4):
        tr = _trainer(51, 400, "meanvar", lr=cfg.LEARNING_RATE)
        drv = LSTMPyDriver(tr, cfg, model_path="/tmp/fov_lstm_drv_%d/LSTM_x.ckpt" % os.getpid(), dropout=0.1, seed=seed)
        drv.fit(FakeDataLayer(61, fixed=True), 6, training_epochs=2, batch_size=10, starting_epoch=3)
        outs.append((tr.flat.clone(), list(drv.history)))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert not torch.equal(outs[0][0], outs[2][0])
    assert len(outs[0][1]) >= 2 and all(np.isfinite(l) for _, l in outs[0][1])
    datadb = {0: {"x": np.zeros((48, 1800))}, 1: {"x": np.zeros((48, 3600))}, 2: {"x": np.zeros((48, 900))}}
    cfg.test_video_ind = 2
    assert total_batch_of(datadb, cfg) == int((5400 - 300) / 10 / 32) * 48


@pytest.mark.parametrize("kind", ["meanvar", "gmm", "raw"])
def test_test_loop_rolls_forward_from_one_second(kind, tmp_path):
    """lstm.py:663-828: zero state per trial, only the LAST second of the window is fed, predict_step seconds rolled forward by
    the trainer's rollout - against the rollout called directly with the same random draws."""
    from longterm360fov_amd.config import default_config
    from longterm360fov_amd.lstm_driver import LSTMPyDriver
    cfg = default_config()
    tr = _trainer(71, 400, kind)
    drv = LSTMPyDriver(tr, cfg, model_path=str(tmp_path / "LSTM_e.ckpt"), dropout=0.0)
    B, P = 8, 4
    outs, gts = drv.test(FakeDataLayer(81), num_trials=2, predict_step=P, batch_size=B, seed=9)
    assert len(outs) == 2 and gts[0].shape == (B, 10, F)
    data = FakeDataLayer(81)
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    for trial in range(2):
        bx = data._get_next_minibatch(None, B)[0]
        x = dev(bx)[:, -1:, :].contiguous()
        z0 = torch.zeros((2, 2, B, 400), device="cuda")
        if kind == "meanvar":
            noise = torch.randn((P, B, F), generator=gen, device="cuda")
            mus, vs, _ = tr.rollout(x, z0, noise)
            ref = torch.cat([mus, vs], 2)
            assert outs[trial].shape == (P, B, 6) and (outs[trial][:, :, 3:] > 0).all()
        elif kind == "gmm":
            u = torch.rand((P, B, FPS), generator=gen, device="cuda")
            zz = torch.randn((P, B, FPS, 3), generator=gen, device="cuda")
            ref, _ = tr.rollout_gmm(x, z0, u, zz)
        else:
            ref, _ = tr.rollout_raw(x, z0, P)
            assert np.abs(outs[trial]).max() <= 1.0
        np.testing.assert_array_equal(outs[trial], ref.cpu().numpy())


def test_stack2_knob_toggled_between_steps_of_one_trainer():
    """Round-4 advisor finding: the trainer cached fov_lstm_stack2_bwd_supported per shape; with FOV_NO_STACK2 flipped after the
    first step the cached 'yes' sent the next step into a launch that refuses.  The answer is asked per step now: steps with the
    knob on / off / on run, and agree with an undisturbed trainer to the two paths' rounding."""
    x, y, init = _batch(91, 32, 10)
    a = (dev(x), dev(y), dev(init))
    ref = _trainer(93, 400, "meanvar")
    tr = _trainer(93, 400, "meanvar")
    for i in range(3):
        ref.train_step(*a)
        if i == 1:
            os.environ["FOV_NO_STACK2"] = "1"
        try:
            tr.train_step(*a)
        finally:
            os.environ.pop("FOV_NO_STACK2", None)
    tr.check(); ref.check()
    d = (tr.flat - ref.flat).abs().max().item()
    assert d <= 2e-5, d


@pytest.mark.parametrize("kind,with_masks", [("meanvar", False), ("gmm", False), ("meanvar", True), ("raw", False)])
def test_carried_state_views_equal_copied_states(kind, with_masks):
    """train_step(..., state_view=True): the final state is written by the kernels straight into one of two carried-state
    buffers and fed back as it is (no padding copy in front of a step, no gather behind it - two launches of lstm.py's step):
    five steps with the state carried as views equal five steps with copied states bit for bit, the views alternate between the
    two buffers, and a view is only overwritten by the call after next.  ('raw' re-runs the stack per window: copies, as before.)"""
    x, y, init = _batch(61, 32, 10, Ty=(2 if kind == "raw" else 1))
    a, b = _trainer(63, 400, kind), _trainer(63, 400, kind)
    rng = np.random.default_rng(7)
    sa = sb = dev(init)
    ptrs = []
    for i in range(5):
        masks = [dev((rng.random((32, 10, 400)) < 0.9) / 0.9), None] if with_masks else None
        la, sa = a.train_step(dev(x), dev(y), sa, masks=masks)
        lb, sb_new = b.train_step(dev(x), dev(y), sb, masks=masks, state_view=True)
        if i > 0 and kind != "raw":
            assert torch.equal(sb, sa_prev)      # the view handed in is still intact after the step that read it
        sb, sa_prev = sb_new, sa
        assert torch.equal(la, lb) and torch.equal(sa, sb) and sb.shape == (2, 2, 32, 400)
        ptrs.append(sb.data_ptr())
    a.check(); b.check()
    assert torch.equal(a.flat, b.flat) and torch.equal(a.ms, b.ms)
    if kind != "raw":
        assert len(set(ptrs)) == 2 and ptrs[0] == ptrs[2] == ptrs[4] and ptrs[1] == ptrs[3] and not sb.is_contiguous()
