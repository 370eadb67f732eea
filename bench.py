#!/usr/bin/env python3
"""Benchmark of the hot path: BASELINE.json metric "sequences/sec (batch=1024, T_in=30 ->
T_out=30, h=256)", configs[1]: target-only seq2seq (mycode/FoV_seq2seq.py), fp32, inference =
encoder + autoregressive decoder, on synthetic trajectories.

A "step" is one pass of the fused path (fov_seq2seq_decode_fwd: encoder launch + decoder launch)
over one resident batch of 1024 sequences per GPU.  N > 1 ranks (torchrun) are independent
replicas on their own batch - the path shards by sequence with no data-path collective
("scaling": "weak"); torch.distributed (RCCL) is only used for the barrier and the max-over-ranks
of the timed region.

Prints ONE JSON line (rank 0).  `roofline` prices the step's two persistent-kernel launches
together against the fp32 MFMA peak, with the duration taken from HIP events on the launch
stream over the timed region; `kernels` breaks it down per launch (event-timed separately).
`cpu_baseline` times the C oracle (oracle/lstm_ref.c, "port") on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (spec)
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 20 --warmup 3`, mean per dispatch:
# encoder 13150 + 6152 KiB, decoder 7441 + 4824 KiB (raw counter values; the kernel's global traffic is 4- and
# 8-byte accesses, for which MI355X_MICROARCH.md gives no correction factor)
PMC_TRAFFIC_CONFIG = (1024, 30, 30, 256, "sigmoid", "auto")
PMC_TRAFFIC_BYTES = (13150 + 6152 + 7441 + 4824) * 1024
PMC_TRAFFIC_SOURCE = "profiles/r01_pmc_v8_summary.csv"


def flops_per_seq(T_in, T_out, F_enc, F_dec, H):
    """Algorithmic forward FLOPs: LSTM step 2(F+H)4H, Dense 2*H*F_dec."""
    enc = T_in * 2 * (F_enc + H) * 4 * H
    dec = T_out * (2 * (F_dec + H) * 4 * H + 2 * H * F_dec)
    return enc, dec


def event_time_ms(fn, iters):
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def bench_train(args, rank, world, use_dist):
    """Secondary measurement: training throughput of the same model/shape (not the BASELINE metric)."""
    import torch.distributed as dist
    from longterm360fov_amd.training import Seq2SeqTrainer
    from oracle import fov_oracle as O
    B, T_in, T_out, H = args.batch, args.t_in, args.t_out, args.hidden
    w = O.init_seq2seq(1234, 90, 6, H, bias_noise=0.05)
    enc, dec0, tgt = O.synthetic_batch(1234 + rank, B, T_in, T_out)
    dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_enc, d_dec, d_tgt = d(enc), d(dec_in), d(tgt)
    tr = Seq2SeqTrainer(w, act=args.act, impl=args.impl)
    for _ in range(args.warmup):
        tr.train_step(d_enc, d_dec, d_tgt, n_global=B * world)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.train_step(d_enc, d_dec, d_tgt, n_global=B * world)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tr.ws.check()
    if rank == 0:
        f_enc, f_dec = flops_per_seq(T_in, T_out, 90, 6, H)
        flop_step = 3 * (f_enc + f_dec) * B      # training step counted as 3x forward
        ms = elapsed / args.steps * 1e3
        print(json.dumps({
            "metric": "training sequences/sec (batch=%d per GPU, T_in=%d->T_out=%d, h=%d)" % (B, T_in, T_out, H),
            "value": world * B * args.steps / elapsed, "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "final_loss": float(loss.item()),
            "config": {"workload": "teacher-forced training step of configs[1] shape (fwd + BPTT + Keras Adam), fp32",
                       "global_batch": B * world, "parallelism": "dp%d, one flat-buffer all-reduce per step" % world},
            "roofline": {"bound": "mfma", "achieved": flop_step / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": flop_step / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "note": "whole step, 3x forward FLOPs"},
            "cpu_baseline": None}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_train_mixing(args, rank, world, use_dist):
    """Secondary measurement, BASELINE.json configs[2]: target+others mixing, 2+2 layers, H=256, T 10->10,
    global batch 4096 sharded over the ranks (512 per GPU at 8 GPUs), one gradient all-reduce per step."""
    import torch.distributed as dist
    from longterm360fov_amd.training import OthersMixingTrainer
    from oracle import fov_oracle as O
    H, T_in, T_out, U = args.hidden, 10, 10, 34
    B = args.batch if args.batch != 1024 else 512
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234 + rank, B, T_in, T_out, num_others=U - 1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a_enc, a_oth, a_dec, a_tgt = d(enc), d(oth), d(dec0), d(tgt)
    tr = OthersMixingTrainer(w, act=args.act, impl=args.impl)
    for _ in range(args.warmup):
        tr.train_step(a_enc, a_oth, a_dec, a_tgt, n_global=B * world)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.train_step(a_enc, a_oth, a_dec, a_tgt, n_global=B * world)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tr.ws.check()
    tr.bwd_scratch.check()
    if rank == 0:
        fwd = (T_in * (2 * (90 + H) * 4 * H + 2 * (H + H) * 4 * H) +
               T_out * (2 * (6 + H) * 4 * H + 2 * (H + H) * 4 * H + 2 * H * 6 + 2 * U * 6 * 6))
        ms = elapsed / args.steps * 1e3
        print(json.dumps({
            "metric": "training sequences/sec, others-mixing 2+2 layers (batch=%d per GPU, T 10->10, h=%d, U=%d)" % (B, H, U),
            "value": world * B * args.steps / elapsed, "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "final_loss": float(loss.item()),
            "config": {"workload": "configs[2] shape: given_others_gt_mean_var_seq2seq training step (fused decoder forward "
                                   "and backward launches)", "global_batch": B * world,
                       "parallelism": "dp%d, one flat-buffer all-reduce per step" % world},
            "roofline": {"bound": "mfma", "achieved": 3 * fwd * B / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": 3 * fwd * B / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "note": "whole step, 3x forward FLOPs"},
            "cpu_baseline": None}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_infer_mixing(args, rank, world, use_dist):
    """Secondary measurement, BASELINE.json configs[2] shape, inference: 2+2-layer others-mixing model, encoder
    over T_in steps + autoregressive decoder with the mixing head, device-resident inputs, replicas only."""
    import torch.distributed as dist
    from longterm360fov_amd.models import OthersMixingSeq2Seq
    from oracle import fov_oracle as O
    H, T_in, T_out, U = args.hidden, 10, 10, 34
    B = args.batch if args.batch != 1024 else 512
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234 + rank, B, T_in, T_out, num_others=U - 1)
    m = OthersMixingSeq2Seq(latent_dim=H, num_user=U, recurrent_activation=args.act, impl=args.impl)
    from longterm360fov_amd.models import _MIX_ORDER
    m.set_weights([w[k] for k in _MIX_ORDER])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a_enc, a_oth, a_dec = d(enc), d(oth), d(dec0)
    for _ in range(args.warmup):
        out = m.predict_device(a_enc, a_oth, a_dec)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = m.predict_device(a_enc, a_oth, a_dec)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ref = O.others_mixing_forward(enc[:32].astype(np.float64), oth[:32].astype(np.float64), dec0[:32].astype(np.float64),
                                      {k: v.astype(np.float64) for k, v in w.items()}, act=args.act)
        err = float(np.abs(out[:32].cpu().numpy() - ref).max())
        fwd = (T_in * (2 * (90 + H) * 4 * H + 2 * (H + H) * 4 * H) +
               T_out * (2 * (6 + H) * 4 * H + 2 * (H + H) * 4 * H + 2 * H * 6 + 2 * U * 6 * 6))
        ms = elapsed / args.steps * 1e3
        print(json.dumps({
            "metric": "sequences/sec, others-mixing 2+2 layers inference (batch=%d per GPU, T 10->10, h=%d, U=%d)" % (B, H, U),
            "value": world * B * args.steps / elapsed, "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2] shape: given_others_gt_mean_var_seq2seq inference (encoder + unrolled "
                                   "no-teacher-forcing decoder with others mixing)", "global_batch": B * world,
                       "parallelism": "replicas x%d (no collective)" % world},
            "roofline": {"bound": "mfma", "achieved": fwd * B / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": fwd * B / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": None},
            "parity": {"max_abs_err_vs_oracle": err, "sequences_checked": 32},
            "cpu_baseline": None}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--t-in", type=int, default=30)
    ap.add_argument("--t-out", type=int, default=30)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--impl", default="auto", choices=["auto", "cluster", "generic"])
    ap.add_argument("--act", default="sigmoid", choices=["sigmoid", "hard_sigmoid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path "
                         "with several ranks on ONE GPU)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "train_mixing", "infer_mixing"],
                    help="infer (default, the BASELINE metric): encoder + autoregressive decoder; train: one "
                         "teacher-forced training step (fwd + BPTT + Adam, data-parallel all-reduce when N > 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1
    device_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")

    from longterm360fov_amd import ops
    from oracle import fov_oracle as O   # synthetic data + Keras initialisers (test infrastructure)
    if args.mode == "train":
        return bench_train(args, rank, world, use_dist)
    if args.mode == "train_mixing":
        return bench_train_mixing(args, rank, world, use_dist)
    if args.mode == "infer_mixing":
        return bench_infer_mixing(args, rank, world, use_dist)

    B, T_in, T_out, H = args.batch, args.t_in, args.t_out, args.hidden
    F_enc, F_dec = 90, 6
    w = O.init_seq2seq(1234, F_enc, F_dec, H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(1234 + rank, B, T_in, T_out)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    d_enc, d_dec0 = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
    out = torch.empty((B, T_out, F_dec), dtype=torch.float32, device="cuda")
    ws = ops.Workspace()

    def step():
        ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, act=args.act, impl=args.impl, workspace=ws, out=out)

    for _ in range(args.warmup):
        step()
    ws.check()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    ws.check()
    step_ms_events = ev0.elapsed_time(ev1) / args.steps
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        f_enc, f_dec = flops_per_seq(T_in, T_out, F_enc, F_dec, H)
        flop_step = (f_enc + f_dec) * B
        achieved = flop_step / (step_ms_events * 1e-3) / 1e12
        # per-launch breakdown: the same two kernels, event-timed on their own
        hT = torch.empty((B, H), dtype=torch.float32, device="cuda")

        def enc_only():
            ops.lstm_seq(d_enc, dw["enc_K"], dw["enc_R"], dw["enc_b"], act=args.act, impl=args.impl,
                         return_sequences=False, workspace=ws)

        enc_only()
        enc_ms = event_time_ms(enc_only, max(5, args.steps // 2))
        dec_ms = max(step_ms_events - enc_ms, 0.0)

        # parity spot check on the measured configuration (first 64 sequences vs the C oracle)
        from oracle import c_oracle as C
        nchk = min(64, B)
        ref = C.seq2seq_decode(enc[:nchk], dec0[:nchk], w, T_out, ops.act_code(args.act))
        got = out[:nchk].cpu().numpy()
        max_abs = float(np.abs(got - ref).max()) if nchk else 0.0
        mse = float(np.mean((got.astype(np.float64) - ref) ** 2)) if nchk else 0.0

        cpu = None
        if not args.no_cpu_baseline:
            reps = 3
            C.seq2seq_decode(enc[:64], dec0[:64], w, T_out)            # warm
            tc = time.perf_counter()
            for _ in range(reps):
                C.seq2seq_decode(enc, dec0, w, T_out, ops.act_code(args.act))
            dt = time.perf_counter() - tc
            cpu = {"value": B * reps / dt, "unit": "sequences/s", "cores": C.num_threads(), "kind": "port",
                   "sample": "%d passes of the C oracle (oracle/lstm_ref.c, OpenMP) over the same %d-sequence "
                             "batch, T %d->%d, H=%d" % (reps, B, T_in, T_out, H)}

        value = world * B * args.steps / elapsed
        # HBM-side bytes per step are not measurable from inside this process: they come from separate rocprofv3
        # --pmc passes over this same command (FETCH_SIZE and WRITE_SIZE, raw KiB per dispatch, encoder +
        # decoder), committed under profiles/; quoted only for the configuration that was profiled.
        traffic, traffic_src = None, None
        if (B, T_in, T_out, H, args.act, args.impl) == PMC_TRAFFIC_CONFIG:
            traffic, traffic_src = PMC_TRAFFIC_BYTES, PMC_TRAFFIC_SOURCE
        result = {
            "metric": "sequences/sec (batch=1024, T_in=30->T_out=30, h=256)",
            "value": value, "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: target-only seq2seq inference (encoder + autoregressive decoder), "
                                   "H=%d, batch=%d per GPU, T_in=%d->T_out=%d, F_enc=90, F_dec=6, fp32, act=%s, impl=%s"
                                   % (H, B, T_in, T_out, args.act, args.impl),
                       "global_batch": B * world, "parallelism": "replicas x%d (no collective)" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "flop_per_launch": flop_step, "launch_ms": step_ms_events,
                         "note": "launch = one step = encoder kernel + decoder kernel of lstm_cluster_kernel"},
            "kernels": {"encoder_ms": enc_ms, "decoder_ms": dec_ms,
                        "encoder_tflops": f_enc * B / (enc_ms * 1e-3) / 1e12,
                        "decoder_tflops": (f_dec * B / (dec_ms * 1e-3) / 1e12) if dec_ms > 0 else None},
            "parity": {"max_abs_err_vs_oracle": max_abs, "mse_vs_oracle": mse, "sequences_checked": nchk},
            "cpu_baseline": cpu,
        }
        if cpu:
            result["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
