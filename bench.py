#!/usr/bin/env python3
"""Headline benchmark: decoded information Gbit/s at a fixed iteration count.

Workload (BASELINE.json configs[1]): regular (3,6) n=4096 rate-1/2 code, 50 flooding
sum-product iterations with early termination off, a 1,048,576-frame batch per GPU of
BPSK/AWGN channel LLRs at Eb/N0 = 1.0 dB generated on the device (Philox) before the timed
region.  One "step" = decode the whole batch (layout change in, message passing, hard
decisions out), count errors against the all-zero codeword and, with more than one GPU,
all-reduce the counter vector.

    python bench.py --gpus N --steps K --warmup W

prints one JSON line on rank 0.  For N > 1 launch one rank per GPU with torch.distributed.run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# Ceilings of the frame-resident kernel, whose messages never leave the CU.  Unit of work: one edge-iteration
# = one message through a check-node and a variable-node update.
SIMDS = 256 * 4            # compute units x SIMD-32 units
CLOCK_HZ = 2.4e9           # maximum shader clock; under this load the chip holds less, so fractions of these
                           # ceilings are lower bounds
# Vector-issue cycles one SIMD needs for 64 edge-iterations (one per lane), at the least, priced with the issue
# rates measured on this chip with every SIMD saturated (tools/valu_rates.hip, profiles/r02/r02_valu_issue_rates.txt:
# v_exp / v_rcp / v_log 8.1 SIMD cycles per wave64 instruction, a packed fp32 instruction 4.45 = 2.225 per
# element, a plain fp32 instruction 2.3 - 2.5; the costs of a mix add up):
#   4 transcendentals per edge (2^|beta|, the reciprocal that gives q, 1/Q, log2);
#   12 fp32 operations per edge.  Check node: scaling of 2^|beta| (1), fma to the denominator of q (1), prefix
#   step 1 - Q and fma (2), suffix step likewise (2), fma joining prefix and suffix (1), fma(1/Q, 2, -1) (1),
#   product of all signs, own sign out of it, sign onto the magnitude (3); variable node: prefix add, suffix
#   add, their sum (3); that is 14, of which the first prefix step and the last suffix step of a node are not
#   needed: 12 on average.
TRANS_CYCLES, FP32_ELEMENT_CYCLES = 8.1, 2.225
VALU_FLOOR_CYCLES_PER_64_EDGE_ITERS = 4 * TRANS_CYCLES + 12 * FP32_ELEMENT_CYCLES
# LDS cycles of one CU per edge-iteration at the data-path rates of the access shapes used: check side
# ds_read_b128 (256 B/clk) and ds_write_b128 (79 B/clk), variable side ds_read_b32 (128 B/clk) and
# ds_write_b32 (64 B/clk), 4 bytes each
LDS_FLOOR_CYCLES_PER_EDGE_ITER = 4 / 256 + 4 / 79 + 4 / 128 + 4 / 64


def resident_ceilings():
    """edge-iterations per second the whole chip could issue / move through LDS at the maximum clock"""
    valu = SIMDS * CLOCK_HZ * 64 / VALU_FLOOR_CYCLES_PER_64_EDGE_ITERS
    lds = 256 * CLOCK_HZ / LDS_FLOOR_CYCLES_PER_EDGE_ITER
    return valu, lds


def transcendental_ceiling():
    """The part of the VALU ceiling that does not depend on how this library arranges the rest of the rule: any
    evaluation of 2 atanh(prod tanh(beta/2)) in LLR form issues at least 4 transcendental instructions per
    edge-iteration on this instruction set (an exponential and a reciprocal to get from beta to a factor, a
    reciprocal and a logarithm to get back), each 8.1 SIMD cycles per wave64; the other vector work priced at zero."""
    return SIMDS * CLOCK_HZ * 64 / (4 * TRANS_CYCLES)


def resident_roofline(edge_iters_per_s):
    valu_peak, lds_peak = resident_ceilings()
    trans_peak = transcendental_ceiling()
    return {
        "bound": "valu_issue", "achieved": edge_iters_per_s / 1e12, "peak": valu_peak / 1e12, "unit": "Tedge-iterations/s",
        "frac": edge_iters_per_s / valu_peak,
        "peak_definition": "1024 SIMDs x 2.4 GHz x 64 lanes / %.1f issue cycles per 64 edge-iterations "
                           "(4 transcendentals x %.1f cycles + 12 fp32 operations x %.3f cycles, the issue rates "
                           "measured by tools/valu_rates.hip); the 12 operations are this library's own recipe"
                           % (VALU_FLOOR_CYCLES_PER_64_EDGE_ITERS, TRANS_CYCLES, FP32_ELEMENT_CYCLES),
        "transcendental_floor": {"peak": trans_peak / 1e12, "unit": "Tedge-iterations/s", "frac": edge_iters_per_s / trans_peak,
                                 "peak_definition": "1024 SIMDs x 2.4 GHz x 64 lanes / (4 transcendentals x %.1f cycles): "
                                                    "the formulation-independent part of the ceiling, all other vector "
                                                    "work priced at zero" % TRANS_CYCLES},
        "lds_ceiling": {"peak": lds_peak / 1e12, "unit": "Tedge-iterations/s", "frac": edge_iters_per_s / lds_peak,
                        "peak_definition": "256 CUs x 2.4 GHz / %.3f LDS cycles per edge-iteration (b128 read and "
                                           "write on the check side, b32 read and write on the variable side)"
                                           % LDS_FLOOR_CYCLES_PER_EDGE_ITER},
    }


def other_configs(L, torch, stream):
    """The other BASELINE configurations, one short measurement each (N = 1 only, after the timed region, not part
    of `value`): C1 the reference's own code (LDPC_GallagerIntStreamSimu.java:20-24) at 1,048,576 frames x 50 fixed
    iterations; C4 the NR-BG1-sized Z = 384 code, 65,536 frames x 50 fixed iterations; C5 the n = 64,800 code with
    1 and 64 frames and early termination.  Each decode runs twice, the second is reported: wall time between
    synchronisations and the message-passing launches between HIP events."""
    out = {}

    def run(dec, ebn0, F, iters, early, seed):
        n = dec.n
        d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        dec.gen_awgn_llr_dev(ebn0, seed, 0, F, d_llr.data_ptr(), stream)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr(), stream)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            _, ms, _ = dec.last_decode_timing()
        mean_it = float(d_it.float().mean())
        fer = float(d_hard.any(dim=1).float().mean())
        del d_llr, d_hard, d_it, d_sw
        return wall, ms / 1e3, mean_it, fer

    def line(dec, F, iters, wall, launch_s, mean_it, fer, workload):
        k = dec.n - dec.m
        d = {"workload": workload, "value": F * k / wall / 1e9, "unit": "Gbit/s", "wall_ms": wall * 1e3, "launch_ms": launch_s * 1e3,
             "kernel": dec.last_decode_kernel(), "mean_iterations": mean_it, "frame_error_rate": fer}
        return d

    # C1
    m, n, rows, cols = L.codes.gallager(1024, 8, 2, seed=1)
    dec = L.Decoder(m, n, rows, cols)
    F, iters = 1 << 20, 50
    wall, launch_s, mean_it, fer = run(dec, 3.0, F, iters, False, 11)
    d = line(dec, F, iters, wall, launch_s, mean_it, fer,
             "Gallager (8,2) n=1024 (the reference's default code), %d frames, %d fixed iterations, AWGN 3.0 dB" % (F, iters))
    d["roofline"] = resident_roofline(F * dec.E * iters / launch_s)
    out["c1"] = d
    dec.close()
    # C4
    _, (m, n, rows, cols) = L.codes.nr_bg1_shaped(z=384, seed=1)
    dec = L.Decoder(m, n, rows, cols)
    F, iters = 1 << 16, 50
    wall, launch_s, mean_it, fer = run(dec, 2.0, F, iters, False, 12)
    d = line(dec, F, iters, wall, launch_s, mean_it, fer,
             "NR-BG1-sized Z=384 (n=26112, E=%d), %d frames, %d fixed iterations, AWGN 2.0 dB" % (dec.E, F, iters))
    b = F * algorithmic_bytes_per_frame(n, dec.E, iters)
    d["roofline"] = {"bound": "hbm", "achieved": b / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b / launch_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_decode": b,
                     "note": "all sweep launches of the decode between two HIP events"}
    out["c4"] = d
    wall, launch_s, mean_it, fer = run(dec, 2.0, F, iters, True, 12)
    out["c4_early_exit"] = line(dec, F, iters, wall, launch_s, mean_it, fer, "the same, early termination")
    dec.close()
    # C5
    m, n, rows, cols = L.codes.regular_ldpc(64800, 3, 6, seed=1)
    dec = L.Decoder(m, n, rows, cols)
    for F in (1, 64):
        wall, launch_s, mean_it, fer = run(dec, 1.6, F, 50, True, 13)
        d = line(dec, F, 50, wall, launch_s, mean_it, fer,
                 "regular (3,6) n=64800, %d frame(s), early termination (<= 50 iterations), AWGN 1.6 dB" % F)
        d["latency_ms"] = wall * 1e3
        b = F * algorithmic_bytes_per_frame(n, dec.E, 1) * mean_it
        d["roofline"] = {"bound": "hbm", "achieved": b / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b / launch_s / 1e9 / HBM_PEAK_GBS,
                         "note": "latency-bound: the batch's whole state (%.1f MB) fits in the caches" % (F * dec.E * 4 / 1e6)}
        out["c5_%d_frame%s" % (F, "" if F == 1 else "s")] = d
    dec.close()
    return out


def algorithmic_bytes_per_frame(n, edges, iters, msg_bytes=4):
    """4*E*s + n*s per iteration: the CN sweep reads and writes E messages, the VN sweep reads
    and writes E messages and reads n channel LLRs."""
    return iters * (4 * edges * msg_bytes + n * msg_bytes)


def cpu_baseline(code, ebn0_db, iters, budget_s):
    """The CPU restatement (oracle/, a port of the reference arithmetic on sparse state) timed
    on this host's cores on a bounded sample of the same workload."""
    from oracle import oracle_capi as oc

    m, n, rows, cols = code
    g = oc.Graph(m, n, rows, cols)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, oc.max_threads(), 16))
    probe = 8 * threads
    llr = oc.gen_awgn_llr(m, n, ebn0_db, 0x5EED, 0, probe)
    t0 = time.perf_counter()
    g.decode_llr_batch(llr, iters, False, threads=threads)
    dt = time.perf_counter() - t0
    frames = int(max(probe, min(65536, budget_s / max(dt / probe, 1e-6))))
    frames -= frames % threads
    llr = oc.gen_awgn_llr(m, n, ebn0_db, 0x5EED, 0, frames)
    t0 = time.perf_counter()
    g.decode_llr_batch(llr, iters, False, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": frames * (n - m) / dt / 1e9,
        "unit": "Gbit/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d frames of the same (3,6) n=%d code, %d fixed iterations, double precision, "
                  "sparse state, %d OpenMP threads, %.1f s" % (frames, n, iters, threads, dt),
    }


def cpu_baseline_dense(code, iters, threads):
    """The reference's own cost model on the same code: per-frame dense alpha[m][n] / beta[n][m], adjacency
    scan per call, d(d-1) tanh per check, dense syndrome (LogDecoder.java:17-123 line by line), on a few
    frames of a BSC at e = 0.12 (no frame converges, so every frame runs all iterations)."""
    from oracle import oracle_capi as oc

    import ldpc_simulator_amd as L

    m, n, rows, cols = code
    H = L.check_matrix_io.to_dense(m, n, rows, cols)
    frames = 2 * threads
    rx = oc.gen_bsc(m, n, 1, 0.12, 0x5EED, 0, frames)
    t0 = time.perf_counter()
    _, it, _ = oc.logdecoder_dense_batch(H, rx, 0.12, iters, llr_mode=0, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": frames * (n - m) / dt / 1e9, "unit": "Gbit/s", "cores": threads, "kind": "port",
            "sample": "%d frames, dense line-by-line mode of the reference's decode(), mean %.1f iterations, %.1f s"
                      % (frames, float(it.mean()), dt)}


def measured_traffic(kernel_prefix, frames, iters, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/<round>/*hbm_traffic.json; FETCH_SIZE and WRITE_SIZE collected in separate passes,
    FETCH_SIZE doubled as gfx950 requires), if one exists for exactly this workload."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*hbm_traffic.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        w = d.get("workload", {"frames": 1 << 20, "iters": 50, "n": 4096})
        if d.get("kernel", "").find(kernel_prefix) >= 0 and (w["frames"], w["iters"], w["n"]) == (frames, iters, n):
            best = d["hbm_bytes_per_launch"]
    return best


def measured_issue_rates(kernel_prefix):
    """VALU and LDS busy fractions of the kernel from the newest committed SQ counter summary
    (profiles/<round>/*_sq_counters_*.txt, collected by tools/pmc_lds.sh): SQ_ACTIVE_INST_VALU counts
    4-cycle units per SIMD (1024 SIMDs), SQ_LDS_IDX_ACTIVE LDS cycles per CU (256), GRBM_GUI_ACTIVE is
    summed over the 8 XCDs.  These describe the build and workload that file was collected on (its path is
    returned as `source`), not this run."""
    import glob
    import re
    out = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*_sq_counters_*.txt"))):
        vals = {}
        try:
            for line in open(path):
                m = re.match(r".*(%s).*\s(SQ_\w+|GRBM_\w+)\s+(?:n=\d+\s+)?([0-9.e+]+)\s*$" % kernel_prefix, line)
                if m:
                    vals[m.group(2)] = float(m.group(3))
        except OSError:
            continue
        if {"SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"} <= set(vals):
            cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
            out = {"valu_busy": vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024.0),
                   "lds_busy": vals["SQ_LDS_IDX_ACTIVE"] / (cycles * 256.0), "source": os.path.relpath(path, ROOT)}
    return out


class _PlumbingDecoder:
    """Stand-in for ldpc_simulator_amd.Decoder on CPU tensors, used only by `--plumbing-test` (a world_size-2
    gloo rehearsal of this file's stepping, all-reduce, timing and reporting without a GPU).  It decodes
    nothing: every frame "decodes" to the all-zero word in max_iters iterations, and frame f of the global
    numbering counts as a frame error when f % 7 == 0, so the all-reduced counters are known in closed form."""

    E = 12288

    def __init__(self, m, n, first_frame):
        self.m, self.n, self.first = m, n, first_frame

    def set_option(self, *a):
        pass

    def gen_awgn_llr_dev(self, *a):
        pass

    def decode_batch_dev(self, d_llr, F, iters, early, d_hard, d_it, d_sw, stream):
        time.sleep(0.01 * (1 + self.first % 3))          # ranks take different times

    def count_errors_dev(self, F, d_hard, d_rx, d_it, d_sw, iters, d_cnt, stream):
        f = self.first + np.arange(F)
        d_cnt[0] += F
        d_cnt[1] += int((f % 7 == 0).sum())
        d_cnt[CNT_FIXED_PLUMBING + iters - 1] += F

    def last_decode_timing(self):
        return 1, 10.0 * (1 + self.first % 3), 0

    def last_decode_kernel(self):
        return "plumbing-test"


CNT_FIXED_PLUMBING = 16


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU per step")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ebn0", type=float, default=1.0)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--engine", choices=["auto", "stream"], default="auto",
                    help="auto: frame-resident kernel when the code fits in LDS; stream: force the frame-major HBM kernel")
    ap.add_argument("--no-stream-leg", action="store_true",
                    help="skip the extra single-step measurement of the streaming engine (N=1 only)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="a decoder option for A/B measurements (ldpc_amd_set_option), e.g. --opt lds_frames=2")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the short measurements of the other configurations (C1, C4, C5; N=1 only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank decodes on GPU 0 and the counters are all-reduced over gloo: a rehearsal of the multi-rank "
                         "path (sharding by global frame index, barriers, maximum over ranks, one JSON line) on a one-GPU box; "
                         "its rates mean nothing")
    ap.add_argument("--plumbing-test", action="store_true",
                    help="no GPU: gloo backend, CPU tensors and a stand-in decoder; rehearses the stepping, the "
                         "all-reduce, the timing over ranks and the JSON line (tests/test_abi_and_host.py)")
    args = ap.parse_args(argv)
    plumbing = args.plumbing_test

    # Only the JSON line may appear on stdout: libraries underneath (RCCL prints a version banner at
    # init) write there too, so everything else goes to stderr while the run lasts.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    if plumbing:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if args.rehearse_on_one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    # LDPC_BENCH_FORCE_DIST=1 initialises the process group also for a single rank (to rehearse the RCCL path)
    use_dist = world > 1 or (os.environ.get("LDPC_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist:
        if plumbing or args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev)

    import ldpc_simulator_amd as L

    F, iters = args.frames, args.iters
    if plumbing:
        m, n = args.n // 2, args.n
        code = None
        dec = _PlumbingDecoder(m, n, rank * F)
        stream = 0
    else:
        code = L.codes.regular_ldpc(args.n, 3, 6, seed=1)
        m, n, rows, cols = code
        dec = L.Decoder(m, n, rows, cols, device=local_rank)
        if args.engine == "stream":
            dec.set_option("engine", "stream")
        for kv in args.opt:
            name, value = kv.split("=", 1)
            dec.set_option(name, int(value))
        stream = torch.cuda.current_stream().cuda_stream

    def buf(shape, dtype):
        return torch.empty(shape, dtype=dtype, device=dev) if not plumbing else torch.empty((1,), dtype=dtype)

    d_llr = buf((F, n), torch.float32)
    dec.gen_awgn_llr_dev(args.ebn0, 0x5EED, rank * F, F, d_llr.data_ptr(), stream)  # global frame index
    d_hard = buf((F, n), torch.uint8)
    d_it = buf(F, torch.int32)
    d_sw = buf(F, torch.int32)
    d_cnt = torch.zeros(L.capi.CNT_FIXED + iters, dtype=torch.int64, device=dev)
    sync()

    kernel_ms, kernel_launches = 0.0, 0
    allreduce_s = 0.0      # host-side time of this rank in the counter all-reduce, its wait for the slowest rank included

    def step(timed):
        nonlocal kernel_ms, kernel_launches, allreduce_s
        d_cnt.zero_()
        dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr(), stream)
        dec.count_errors_dev(F, d_hard.data_ptr(), 0, d_it.data_ptr(), d_sw.data_ptr(), iters,
                             d_cnt if plumbing else d_cnt.data_ptr(), stream)
        if use_dist:
            if timed:
                sync()                      # the decode is over: what follows is the exchange alone
                ta = time.perf_counter()
            dist.all_reduce(d_cnt)  # the only exchange: error counters (RCCL)
            if timed:
                sync()
                allreduce_s += time.perf_counter() - ta
        if timed:
            nl, ms, _ = dec.last_decode_timing()  # HIP events on `stream` around the message-passing kernel
            kernel_ms += ms
            kernel_launches += nl

    for _ in range(args.warmup):
        step(False)
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    dt_own_work = time.perf_counter() - t0 - allreduce_s   # this rank's steps without the exchange and its waiting
    if use_dist:
        dist.barrier()
    sync()
    dt_rank = time.perf_counter() - t0
    dt = dt_rank
    per_rank = [[dt_rank, dt_own_work, allreduce_s]]
    if use_dist:
        t = torch.tensor([dt_rank], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_rank, dt_own_work, allreduce_s], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(x) for x in e.cpu()] for e in every]

    kernel = dec.last_decode_kernel()
    cnt = d_cnt.cpu().numpy()
    k = n - m
    total_frames = F * world * args.steps
    gbit = total_frames * k / dt / 1e9
    b_frame = algorithmic_bytes_per_frame(n, dec.E, iters)
    frames_per_launch = F * args.steps / max(kernel_launches, 1)
    launch_s = kernel_ms / 1e3 / max(kernel_launches, 1)
    hbm_achieved = frames_per_launch * b_frame / launch_s / 1e9 if launch_s > 0 else 0.0
    hbm_equivalent = {
        "bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": frames_per_launch * b_frame,
    }
    if kernel == "bp_lds_kernel":
        # The messages of a frame stay in the LDS of one CU for all iterations: the algorithmic message bytes
        # never travel, so the HBM figure is an equivalent, not a fraction of a ceiling.  What bounds this
        # kernel is vector instruction issue; the LDS data path is the next ceiling.
        edge_iters = frames_per_launch * dec.E * iters / launch_s if launch_s > 0 else 0.0
        hbm_equivalent["note"] = ("algorithmic message bytes / time; they stay in LDS, so this exceeds the HBM peak "
                                  "and `traffic` (measured HBM bytes per launch) is far below it")
        roofline = resident_roofline(edge_iters)
        roofline.update({
            "traffic": measured_traffic(kernel, F, iters, n),
            "kernel": kernel, "launch_ms": launch_s * 1e3,
            "edge_iterations_per_launch": frames_per_launch * dec.E * iters,
            "hbm_equivalent": hbm_equivalent,
            "issue_from_committed_profile": measured_issue_rates(kernel),
        })
    else:
        roofline = dict(hbm_equivalent, traffic=measured_traffic(kernel, F, iters, n), kernel=kernel,
                        launch_ms=launch_s * 1e3, note="frame-major streaming kernel: HBM-bound")

    out = {
        "metric": "decoded info Gbit/s at fixed iters",
        "value": gbit,
        "unit": "Gbit/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not plumbing else "plumbing-test (no decoding)",
        "config": {
            "workload": "regular (3,6) n=%d rate-1/2, %d flooding sum-product iterations (early exit off), "
                        "%d frames per GPU per step, BPSK/AWGN Eb/N0=%.1f dB LLRs generated on device (Philox4x32-10), "
                        "all-zero codeword" % (n, iters, F, args.ebn0),
            "frames_per_gpu": F, "iterations": iters, "n": n, "m": m, "edges": int(dec.E),
            "parallelism": "frames sharded, %d rank(s), counters all-reduced" % world
                           + (" (REHEARSAL: all ranks on one GPU, gloo; rates meaningless)" if args.rehearse_on_one_gpu else ""),
        },
        "roofline": roofline,
        "frame_error_rate": float(cnt[1]) / max(float(cnt[0]), 1.0),
        # per rank: wall time of the timed steps, the part spent in its own decode and accounting, and the part
        # spent in the counter all-reduce (which includes waiting for the slowest rank)
        "per_rank": {"ms_per_step": [r[0] / args.steps * 1e3 for r in per_rank],
                     "own_work_ms_per_step": [r[1] / args.steps * 1e3 for r in per_rank],
                     "allreduce_ms_per_step": [r[2] / args.steps * 1e3 for r in per_rank]},
        "counters": {"frames": int(cnt[0]), "frame_errors": int(cnt[1])},
    }
    if world == 1 and kernel == "bp_lds_kernel" and not args.no_stream_leg:
        # the same batch once through the frame-major HBM streaming engine, for the HBM roofline figure
        dec.set_option("engine", "stream")
        dec.set_option("stream_kernel", 1)  # the one-launch streaming kernel: one launch = one profiler row
        for timed in (False, True):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr(), stream)
            nl, ms, _ = dec.last_decode_timing()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        # ... and through the row-parallel sweeps, the streaming engine's default (many launches: their time is the
        # span between the HIP events around all of them)
        dec.set_option("stream_kernel", 0)
        for timed in (False, True):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr(), stream)
            _, ms_sweeps, _ = dec.last_decode_timing()
            torch.cuda.synchronize()
        sweeps_kernel = dec.last_decode_kernel()
        dec.set_option("engine", "auto")
        out["streaming_sweeps"] = {
            "value": F * k / (ms_sweeps / 1e3) / 1e9, "unit": "Gbit/s", "decode_ms": ms_sweeps, "kernel": sweeps_kernel,
            "roofline": {"bound": "hbm", "achieved": F * b_frame / (ms_sweeps / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": F * b_frame / (ms_sweeps / 1e3) / 1e9 / HBM_PEAK_GBS},
        }
        out["streaming_engine"] = {
            "value": F * k / (t2 - t1) / 1e9, "unit": "Gbit/s", "ms_per_step": (t2 - t1) * 1e3,
            "roofline": {"bound": "hbm", "achieved": F * b_frame / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": F * b_frame / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": measured_traffic(dec.last_decode_kernel(), F, iters, n),
                         "kernel": dec.last_decode_kernel(), "launch_ms": ms / max(nl, 1)},
        }
    if world == 1 and not plumbing and not args.no_config_legs and args.n == 4096:
        # free the headline batch first: the C4 leg needs 47 GB of its own
        del d_llr, d_hard, d_it, d_sw
        dec.close()
        torch.cuda.empty_cache()
        out["configs"] = other_configs(L, torch, stream)
    if rank == 0:
        if world == 1 and args.cpu_seconds > 0 and not plumbing:
            out["cpu_baseline"] = cpu_baseline(code, args.ebn0, iters, args.cpu_seconds)
            out["cpu_baseline_dense"] = cpu_baseline_dense(code, iters, out["cpu_baseline"]["cores"])
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
