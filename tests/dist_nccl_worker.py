"""Worker of tests/test_gpu_dist_nccl.py: ONE rank (this box has one GPU) on the real RCCL backend.  FOV_FORCE_DIST=1
makes the trainers take their data-parallel branch at world size 1: tail all-reduce issued async under the encoder's
BPTT, poison slot + head all-reduce, work.wait() on the launch stream, guarded optimizer on the all-reduced poison slot;
parallel.broadcast_index on the nccl backend.  Three steps must leave bit-identical weights to the plain path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from longterm360fov_amd import parallel, training
    from oracle import fov_oracle as O

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    results = {}
    # a4 / configs[2] trainer (fp32 and bf16 operands) and the target-only trainer
    wm = O.init_others_mixing(223, H=256, num_user=5, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(224, 41, 3, 4, num_others=4)
    ws2 = O.init_seq2seq(13, H=128, bias_noise=0.05)
    enc2, dec02, tgt2 = O.synthetic_batch(14, 40, 6, 5)
    dec_in2 = np.concatenate([dec02, tgt2[:, :-1]], axis=1)
    for name, make, batch in (
            ("mixing_f32", lambda: training.OthersMixingTrainer(wm, dtype="f32"), (enc, oth, dec0, tgt)),
            ("mixing_bf16", lambda: training.OthersMixingTrainer(wm, dtype="bf16"), (enc, oth, dec0, tgt)),
            ("seq2seq", lambda: training.Seq2SeqTrainer(ws2), (enc2, dec_in2, tgt2))):
        outs = {}
        for forced in ("0", "1", "1+overlap"):
            os.environ["FOV_FORCE_DIST"] = forced[0]
            assert parallel.dp_active() == (forced != "0")
            tr = make()
            tr.overlap_allreduce = forced.endswith("overlap")   # tail all-reduce under the encoder's BPTT / one all-reduce at the end
            losses = []
            for _ in range(3):
                losses.append(float(tr.train_step(*[dev(a) for a in batch], n_global=batch[0].shape[0]).item()))
            tr.check()
            outs[forced] = (tr.flat.clone(), losses)
        same = all(bool(torch.equal(outs["0"][0], outs[k][0])) and outs["0"][1] == outs[k][1] for k in ("1", "1+overlap"))
        results[name] = same
        print("%s: forced-DP over RCCL == plain path: %s (losses %s)" % (name, same, outs["1"][1]), flush=True)
    os.environ["FOV_FORCE_DIST"] = "1"
    idx = np.random.default_rng(0).permutation(1000)
    got = parallel.broadcast_index(idx)
    results["broadcast_index"] = bool((got == idx).all())
    print("broadcast_index over nccl: %s" % results["broadcast_index"], flush=True)
    t = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    ok = all(results.values())
    print("DIST_NCCL_WORKER %s" % ("OK" if ok else "FAIL"), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
