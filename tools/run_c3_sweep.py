#!/usr/bin/env python3
"""The Eb/N0 sweep of the (3,6) n = 4096 code at its stated workload: Eb/N0 = 0, 0.5, ..., 4.0 dB, early
termination, 50 iterations at most, random codewords (u * encodedG, GenerateC.java:5-27), every point run
until at least 100 frames have failed or the frame budget (1e8 by default, the budget of an error rate of
1e-6) is spent.  It is the loop over error rates of LDPC_GallagerIntStreamSimu.java:79-213 with the
stop rule in place of the fixed numFrames.

    python tools/run_c3_sweep.py [--budget 100000000] [--out profiles/r03/c3_sweep]

With more than one GPU launch one rank per GPU (python -m torch.distributed.run --nproc-per-node N ...):
frames shard by global index, the counters are all-reduced after every batch, the table does not depend
on N.  Writes <out>.json and <out>.md."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def wilson(k, nobs, z=1.959964):
    if nobs == 0:
        return 0.0, 1.0
    ph = k / nobs
    den = 1 + z * z / nobs
    mid = (ph + z * z / (2 * nobs)) / den
    half = z * math.sqrt(ph * (1 - ph) / nobs + z * z / (4 * nobs * nobs)) / den
    return max(0.0, mid - half), min(1.0, mid + half)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=int, default=10 ** 8)
    ap.add_argument("--min-errors", type=int, default=100)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per batch over all ranks (0 = 2^20 per rank, so that a rank's share of a batch does not shrink "
                         "as ranks are added; the stop rule is looked at once per batch, so the frames a point runs depend on it)")
    ap.add_argument("--host-counters", action="store_true",
                    help="the counters of every batch through the host (ldpc_amd_simulate) instead of staying on the device")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--points", default="0,0.5,1,1.5,2,2.5,3,3.5,4")
    ap.add_argument("--seed", type=int, default=0x5EED)
    ap.add_argument("--zero-codeword", action="store_true", help="send the all-zero word instead of u * encodedG")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank on GPU 0, counters all-reduced over gloo: the multi-rank procedure on a one-GPU box "
                         "(with --batch fixed, the counts must equal a one-rank run's; the rates mean nothing)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03", "c3_sweep"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ldpc_simulator_amd as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
    if args.zero_codeword:
        dec = L.Decoder(m, n, rows, cols, device=local_rank)
    else:
        # the decoder works on the column-permuted H that belongs to the systematic generator
        # (TissueEncoder.EncodeH / EncodeG, LDPC_GallagerIntStreamSimu.java:64-76)
        eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
        r2, c2 = np.nonzero(eh)
        dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32), device=local_rank)
        dec.set_generator(eg)
    k = n - m
    batch = args.batch or world << 20
    table = []
    for ebn0 in [float(x) for x in args.points.split(",")]:
        fn = lambda first, count: dec.simulate(L.capi.CH_AWGN, ebn0, args.seed, first, count, args.iters, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # the counters stay on the device: one in-place all-reduce and one word read back per batch
        acc = None if args.host_counters else L.sim.device_accumulator(dec, L.capi.CH_AWGN, ebn0, args.seed, args.iters, True)
        if acc is not None and args.rehearse_on_one_gpu:
            # gloo with device tensors: the leg is waited for before the all-reduce is issued (issued while the legs of
            # two ranks were still running on the one GPU, c10d's gloo path never returned)
            inner = acc

            def acc(first, count, tensor, inner=inner):
                inner(first, count, tensor)
                torch.cuda.synchronize()
        c = L.sim.simulate_until(fn, args.iters, min_frame_errors=args.min_errors, max_frames=args.budget,
                                 batch=batch, rank=rank, world=world, accumulate_fn=acc,
                                 device=None if (args.rehearse_on_one_gpu and acc is None) else dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = L.sim.rates(c, m, n)
        lo, hi = wilson(int(c[1]), int(c[0]))
        row = {
            "ebn0_db": ebn0, "frames": int(c[0]), "frame_errors": int(c[1]), "FER": r["FER"], "FER_ci95": [lo, hi],
            "info_bit_errors": int(c[2]), "IBER": r["IBER"], "undetected": int(c[6]),
            "mean_iters": float((c[8] + c[9]) / max(int(c[0]), 1)), "mean_iters_decoded": r["avg_iters_correct"],
            "channel_BER": r["channel_BER"], "wall_s": dt, "frames_per_s": int(c[0]) / dt,
            "info_gbit_per_s": int(c[0]) * k / dt / 1e9,
            "stopped_by": "frame errors" if int(c[1]) >= args.min_errors else "budget",
            "kernel": dec.last_decode_kernel(),
        }
        assert int(c[16:].sum()) == int(c[0])           # every frame is in the iteration histogram
        assert int(c[6]) <= int(c[1])                   # an undetected error is a frame error
        table.append(row)
        if rank == 0:
            print(json.dumps(row), flush=True)
    if rank == 0:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        meta = {"code": "regular (3,6) n=4096, seed 1" + ("" if args.zero_codeword else ", column-permuted for the systematic generator"),
                "codewords": "all-zero" if args.zero_codeword else "random, u * encodedG (Philox stream 2)",
                "max_iters": args.iters, "early_termination": True, "stop_rule": ">= %d frame errors or %d frames" % (args.min_errors, args.budget),
                "batch": batch, "gpus": world, "seed": args.seed, "device": torch.cuda.get_device_name(local_rank)}
        with open(args.out + ".json", "w") as f:
            json.dump({"meta": meta, "points": table}, f, indent=1)
        with open(args.out + ".md", "w") as f:
            f.write("# Eb/N0 sweep, (3,6) n = 4096, early termination, <= %d iterations, %s codewords, %d GPU(s)\n\n" %
                    (args.iters, meta["codewords"].split(",")[0], world))
            f.write("Stop rule per point: %s; batches of %d frames.  `%s`\n\n" % (meta["stop_rule"], batch, " ".join(sys.argv)))
            f.write("| Eb/N0 dB | frames | frame errors | FER | 95 % interval | IBER | undetected | mean iterations | wall s | M frames/s | info Gbit/s | stopped by |\n")
            f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
            for r in table:
                f.write("| %.1f | %d | %d | %.3e | [%.2e, %.2e] | %.3e | %d | %.2f | %.2f | %.2f | %.2f | %s |\n" % (
                    r["ebn0_db"], r["frames"], r["frame_errors"], r["FER"], r["FER_ci95"][0], r["FER_ci95"][1], r["IBER"],
                    r["undetected"], r["mean_iters"], r["wall_s"], r["frames_per_s"] / 1e6, r["info_gbit_per_s"], r["stopped_by"]))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
