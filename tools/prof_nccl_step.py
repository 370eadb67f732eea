"""configs[2] training step (others-mixing 2+2 layers, 512 sequences per GPU, T 10->10) with the data-parallel branch forced on at
world size 1 over backend "nccl" (= RCCL): what a rocprofv3 kernel trace of this shows is where the collectives' device work
lands relative to the encoder's BPTT.  The program sets its own rendezvous variables, so it is what follows `--` directly:
    rocprofv3 --kernel-trace --stats -d gpurun_out/nccl_prof -- python3 tools/prof_nccl_step.py [--dtype bf16]"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--plain", action="store_true", help="no process group: the non-distributed step, for comparison")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    if not args.plain:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["FOV_FORCE_DIST"] = "1"
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from longterm360fov_amd import training
    from oracle import fov_oracle as O
    B, T_in, T_out, U = 512, 10, 10, 34
    w = O.init_others_mixing(1234, H=256, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    batch = [d(enc), d(oth), d(dec0), d(tgt)]
    tr = training.OthersMixingTrainer(w, dtype=args.dtype)
    for _ in range(5):
        tr.train_step(*batch, n_global=B)
    tr.check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step(*batch, n_global=B)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    tr.check()
    print("%s step %s: %.4f ms per step" % ("plain" if args.plain else "forced-DP over RCCL (world 1)", args.dtype, ms), flush=True)
    if not args.plain:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
