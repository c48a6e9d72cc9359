#!/usr/bin/env python3
"""Fixed cost of one batch of an error-rate point (sim.simulate_until) on one rank, with the process group
initialised so that the all-reduce really runs (RCCL, world 1):

    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 python tools/perf_sim_batch_overhead.py

(3,6) n = 4096 code, AWGN 4 dB, early termination, random codewords: the point of the sweep where a batch is
shortest.  For batches of 2^14 ... 2^20 frames: time per batch with the counters through the host
(ldpc_amd_simulate -> numpy -> tensor -> device -> all-reduce -> host) and kept on the device
(ldpc_amd_simulate_dev -> in-place all-reduce -> one word read back).  The fixed cost per batch is the time per
batch minus frames x the per-frame time of the largest batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    import torch
    import torch.distributed as dist
    import ldpc_simulator_amd as L

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
    eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
    r2, c2 = np.nonzero(eh)
    dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
    dec.set_generator(eg)
    Lmax, ebn0, seed = 50, 4.0, 0x5EED
    fn = lambda first, count: dec.simulate(L.capi.CH_AWGN, ebn0, seed, first, count, Lmax, True)
    acc = L.sim.device_accumulator(dec, L.capi.CH_AWGN, ebn0, seed, Lmax, True)
    rows_out = []
    for lb in (20, 17, 14):
        batch = 1 << lb
        batches = max(4, (1 << 22) >> lb)
        res = {}
        for name, kw in (("host", dict(accumulate_fn=None)), ("device", dict(accumulate_fn=acc))):
            for rep in range(2):      # the first repetition allocates
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                c = L.sim.simulate_until(fn, Lmax, min_frame_errors=10 ** 9, max_frames=batch * batches, batch=batch,
                                         device=dev, always_reduce=True, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            assert int(c[0]) == batch * batches
            res[name] = (dt / batches, c)
        assert (res["host"][1] == res["device"][1]).all()
        rows_out.append((batch, batches, res["host"][0], res["device"][0]))
    x = np.array([r[0] for r in rows_out], float)
    print("batch frames, batches, ms per batch with host counters, ms per batch with device counters")
    for batch, batches, th, td in rows_out:
        print("%8d %5d %9.3f %9.3f" % (batch, batches, th * 1e3, td * 1e3))
    for name, i in (("host", 2), ("device", 3)):
        slope, icpt = np.polyfit(x, np.array([r[i] for r in rows_out]), 1)
        print("%s counters: %.3f ms fixed per batch + %.2f ns per frame" % (name, icpt * 1e3, slope * 1e9))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
