"""Monte-Carlo frames sharded over the GPUs of one node.

Frames are independent (the reference runs them as an unordered parallel stream,
LDPC_GallagerIntStreamSimu.java:99), so rank r of W takes a contiguous range of global
frame indices and no data moves between ranks while decoding.  The only exchange is the
sum of the counter vectors, which replaces the reference's synchronized(lock) block
(LDPC_GallagerIntStreamSimu.java:139-183): one all-reduce of (16 + maxL) integers.
The channel generator is keyed by the global frame index, so the union of what W ranks
decode is the same set of frames for every W.
"""
import numpy as np

from . import capi


def shard_range(first_frame, frames, rank, world):
    """Contiguous share of [first_frame, first_frame + frames) for `rank` of `world`."""
    base, extra = divmod(int(frames), int(world))
    start = first_frame + rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def allreduce_counters(counters, device=None):
    """Sum a uint64 counter vector over the default process group (RCCL on GPUs, gloo on
    CPU).  With no initialised process group the vector is returned unchanged."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return counters
    t = torch.from_numpy(counters.astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def simulate_sharded(simulate_fn, first_frame, frames, max_iters, rank=0, world=1, device=None):
    """Run `simulate_fn(first, count) -> uint64[16 + max_iters]` on this rank's share and
    return the all-reduced counters.  On a GPU `simulate_fn` is Decoder.simulate bound to
    a channel; tests inject a CPU stand-in to exercise the sharding and the reduction."""
    start, count = shard_range(first_frame, frames, rank, world)
    local = np.zeros(capi.CNT_FIXED + int(max_iters), np.uint64)
    if count > 0:
        local = np.asarray(simulate_fn(start, count), dtype=np.uint64)
    return allreduce_counters(local, device)


def simulate_until(simulate_fn, max_iters, min_frame_errors=100, max_frames=10 ** 8, batch=1 << 20, first_frame=0,
                   rank=0, world=1, device=None, accumulate_fn=None, always_reduce=False):
    """One point of an error-rate curve: decode batches of `batch` frames (sharded over the ranks, global
    frame indices first_frame, first_frame + 1, ...) until at least `min_frame_errors` frames have failed or
    `max_frames` have been decoded, whichever comes first.  The counters are all-reduced after every batch,
    so every rank takes the same decision to stop; the frames decoded, and so the result, depend on `batch`
    (the stop rule is looked at once per batch) but not on the number of ranks.  Returns the summed counter
    vector (numpy uint64).

    Two ways to run a rank's share of a batch:
      * simulate_fn(first, count) -> uint64[16 + max_iters] on the host (Decoder.simulate; CPU stand-ins in tests);
      * accumulate_fn(first, count, tensor): ADD the share's counters into `tensor` (int64[16 + max_iters] on
        `device`) without waiting for them (Decoder.simulate_dev, see device_accumulator).  Then the counters
        never leave the device: the batch's vector is all-reduced in place, added to the running total there,
        and one word (the frame errors so far) is read back per batch for the stop rule.
    A caller that wants a rank's share to stay large as ranks are added scales `batch` itself (tools/run_c3_sweep.py
    uses world x 2^20).  The all-reduce is stream-ordered with RCCL (backend "nccl"); with the gloo backend and
    device tensors let accumulate_fn wait for the device before it returns."""
    if accumulate_fn is None:
        total = np.zeros(capi.CNT_FIXED + int(max_iters), np.uint64)
        done = 0
        while done < max_frames:
            count = int(min(batch, max_frames - done))
            total += simulate_sharded(simulate_fn, first_frame + done, count, max_iters, rank, world, device)
            done += count
            if int(total[capi.CNT_FRAME_ERRORS]) >= min_frame_errors:
                break
        return total
    import torch
    import torch.distributed as dist

    reduce = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or always_reduce)
    total = torch.zeros(capi.CNT_FIXED + int(max_iters), dtype=torch.int64, device=device)
    local = torch.zeros_like(total)
    done = 0
    while done < max_frames:
        count = int(min(batch, max_frames - done))
        start, share = shard_range(first_frame + done, count, rank, world)
        local.zero_()
        if share > 0:
            accumulate_fn(start, share, local)
        if reduce:
            dist.all_reduce(local, op=dist.ReduceOp.SUM)
        total += local
        done += count
        if int(total[capi.CNT_FRAME_ERRORS].item()) >= min_frame_errors:     # the one word that crosses to the host
            break
    return total.cpu().numpy().astype(np.uint64)


def device_accumulator(dec, channel, param, seed, max_iters, early_term=True):
    """accumulate_fn for simulate_until on a GPU: a rank's share of a batch through ldpc_amd_simulate_dev on the
    current torch stream, counters added into the given device tensor."""
    import torch

    def accumulate(first, count, tensor):
        assert tensor.dtype == torch.int64 and tensor.is_cuda and tensor.numel() >= capi.CNT_FIXED + int(max_iters)
        dec.simulate_dev(channel, param, seed, first, count, max_iters, early_term, tensor.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    return accumulate


def rates(counters, m, n):
    """FER / IFER / CFER / IBER / mean iterations as LDPC_GallagerIntStreamSimu.java:186-212
    derives them from the counters."""
    c = np.asarray(counters, dtype=np.float64)
    frames = c[capi.CNT_FRAMES]
    k = n - m
    nan = float("nan")
    ok = frames - c[capi.CNT_FRAME_ERRORS]
    return {
        "frames": int(frames),
        "FER": c[capi.CNT_FRAME_ERRORS] / frames if frames else nan,
        "IFER": c[capi.CNT_INFO_FRAME_ERRORS] / frames if frames else nan,
        "CFER": c[capi.CNT_CHECK_FRAME_ERRORS] / frames if frames else nan,
        "IBER": c[capi.CNT_INFO_BIT_ERRORS] / (k * frames) if frames and k else nan,
        "undetected": int(c[capi.CNT_UNDETECTED]),
        "miscorrected_bits": int(c[capi.CNT_MISCORRECTED]),
        "avg_iters_correct": c[capi.CNT_ITERS_CORRECT] / ok if ok else nan,
        "avg_iters_wrong": c[capi.CNT_ITERS_WRONG] / c[capi.CNT_FRAME_ERRORS] if c[capi.CNT_FRAME_ERRORS] else nan,
        "channel_BER": c[capi.CNT_CHANNEL_ERRORS] / (k * frames) if frames and k else nan,
    }
