"""CPU-side checks: the C ABI library loads and exports every symbol the header declares,
compute entry points fail loudly without a GPU, the H text format and the code builders
behave as the reference's, and the N > 1 path (frame sharding + counter all-reduce) works
with world_size 2 on gloo."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ldpc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldpc_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.capi.load_library()
    names = _declared_functions()
    assert len(names) >= 17
    for name in names:
        assert hasattr(lib, name), "libldpc_amd.so does not export %s" % name
    assert sorted(pkg.capi.SIGNATURES) == names, "capi.SIGNATURES and include/ldpc_amd.h disagree"
    header = open(os.path.join(ROOT, "include", "ldpc_amd.h")).read()
    assert lib.ldpc_amd_version() == int(re.search(r"#define LDPC_AMD_VERSION (\d+)", header).group(1))
    # every option key of the header is known to the binding under the same number
    keys = dict(re.findall(r"#define LDPC_AMD_OPT_([A-Z_]+) (\d+)", header))
    assert {k.lower().replace("tune_", ""): int(v) for k, v in keys.items()} == pkg.capi.OPT


def test_product_library_reads_no_environment_while_decoding():
    """Engine choice is a per-context option; the only getenv in the library is the one in ldpc_amd_create."""
    csrc = os.path.join(ROOT, "ldpc-simulator_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                if "getenv" in line:
                    hits.append((f, i, line.strip()))
    assert len(hits) == 1 and "LDPC_AMD_VERBOSE" in hits[0][2], hits


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.LdpcError) as ei:
        pkg.Decoder(1, 3, [0, 0, 0], [0, 1, 2])
    assert ei.value.code == pkg.capi.ENODEV
    with pytest.raises(pkg.LdpcError):
        pkg.LogDecoder.decode([[1, 1, 1]], [0, 0, 0], 0.1, 5)


def test_product_code_never_touches_the_oracle():
    pkgdir = os.path.join(ROOT, "ldpc-simulator_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % f


def test_check_matrix_text_format(tmp_path, pkg):
    io = pkg.check_matrix_io
    H = np.array([[1, 0, 1, 0], [0, 1, 1, 0], [0, 0, 0, 1]], np.int32)
    p = tmp_path / "h.txt"
    io.save_dense(str(p), H)
    # CheckMatrixIO.java:13-19: "m n" line, then "i j " pairs row-major, each followed by a space
    assert p.read_text() == "3 4\n0 0 0 2 1 1 1 2 2 3 "
    m, n, rows, cols = io.load_check_matrix(str(p))
    assert (io.to_dense(m, n, rows, cols) == H).all()
    # out-of-range pairs are ignored, a trailing unpaired integer is dropped, repeats collapse
    # (CheckMatrixIO.java:37-45)
    q = tmp_path / "q.txt"
    q.write_text("2 3\n0 0 0 0 5 1 1 7 1 2 1")
    m, n, rows, cols = io.load_check_matrix(str(q))
    assert (m, n) == (2, 3) and list(zip(rows.tolist(), cols.tolist())) == [(0, 0), (1, 2)]
    e = tmp_path / "e.txt"
    e.write_text("")
    assert io.load_check_matrix(str(e)) is None  # CheckMatrixIO.java:30


def test_code_builders(pkg):
    codes = pkg.codes
    m, n, rows, cols = codes.gallager(1024, 8, 2, seed=1)
    dc, dv = codes.degrees(m, n, rows, cols)
    assert (m, n, rows.size) == (256, 1024, 2048) and (dc == 8).all() and (dv == 2).all()
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    # block 0 of GenerateMatrix.java:17-21: row i covers columns [8i, 8i+8)
    for i in range(128):
        assert H[i].nonzero()[0].tolist() == list(range(8 * i, 8 * i + 8))
    # every block has exactly one 1 per column (GenerateMatrix.java:25-32)
    assert (H[:128].sum(axis=0) == 1).all() and (H[128:].sum(axis=0) == 1).all()
    with pytest.raises(ValueError):
        codes.gallager(4096, 6, 3, seed=1)  # 4096 % 6 != 0, Divisibility.java:4-7
    m, n, rows, cols = codes.regular_ldpc(4096, 3, 6, seed=1)
    dc, dv = codes.degrees(m, n, rows, cols)
    assert (m, rows.size) == (2048, 12288) and (dc == 6).all() and (dv == 3).all()
    assert (np.diff(rows.astype(np.int64) * n + cols) > 0).all()  # row-major, no repeats
    base, (m, n, rows, cols) = codes.nr_bg1_shaped(z=384, seed=1)
    dc, dv = codes.degrees(m, n, rows, cols)
    assert (m, n, rows.size) == (17664, 26112, 121344)
    assert dc.min() == 3 and dc.max() == 19 and dv.min() == 1 and dv.max() == 30
    # QC expansion rule of GenerateMatrix.java:139-150
    b = np.array([[2, -1], [0, 1]])
    m, n, rows, cols = codes.qc_expand(b, 3)
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    assert H[0, 2] == 1 and H[1, 0] == 1 and H[2, 1] == 1 and H[:3, 3:].sum() == 0
    assert H[3, 0] == 1 and H[3, 4] == 1 and H[5, 3] == 1
    base, _ = codes.random_qc(108, 5, 10, seed=3)
    assert not codes.has_base_4cycle(base, 108)
    assert ((base >= 0).sum(axis=0) >= 2).all() and ((base >= 0).sum(axis=1) >= 2).all()


def test_shard_ranges_partition_the_frames(pkg):
    for frames, world in ((10, 3), (1 << 20, 8), (5, 8), (0, 2)):
        got = [pkg.sim.shard_range(100, frames, r, world) for r in range(world)]
        assert sum(c for _, c in got) == frames
        pos = 100
        for start, count in got:
            assert start == pos
            pos += count


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
import ldpc_simulator_amd as L
from oracle import oracle_capi as oc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
m, n, rows, cols = L.codes.gallager(96, 8, 2, seed=3)
g = oc.Graph(m, n, rows, cols)
# CPU stand-in for Decoder.simulate so that the sharding and the reduction run without a GPU
fn = lambda first, count: g.simulate(oc.CH_BSC_ALL, 0.04, 5, first, count, 20, threads=1)
total = L.sim.simulate_sharded(fn, 1000, 301, 20, rank, world)
whole = g.simulate(oc.CH_BSC_ALL, 0.04, 5, 1000, 301, 20, threads=1)
assert (total == whole).all(), (total, whole)
assert total[0] == 301
# stop rule of an error-rate point: same batches, same stop on every rank, same result as one rank
calls = []
def fn2(first, count):
    calls.append((first, count))
    return fn(first, count)
until = L.sim.simulate_until(fn2, 20, min_frame_errors=12, max_frames=2000, batch=150, first_frame=40, rank=rank, world=world)
alone, done = np.zeros_like(until), 0
while done < 2000 and alone[1] < 12:
    alone += g.simulate(oc.CH_BSC_ALL, 0.04, 5, 40 + done, 150, 20, threads=1)
    done += 150
assert (until == alone).all() and until[1] >= 12 and until[0] == done < 2000, (until[:3], alone[:3])
assert len(calls) == done // 150 and sum(c for _, c in calls) == L.sim.shard_range(0, 150, rank, world)[1] * (done // 150)
capped = L.sim.simulate_until(fn, 20, min_frame_errors=10 ** 9, max_frames=260, batch=100, rank=rank, world=world)
assert capped[0] == 260
# the same point with the counters kept in a tensor (the stand-in for Decoder.simulate_dev adds into it): one
# all-reduce of that tensor per batch, one word read back; same stop batch, same counters, for every world size
import torch
shares = []
def acc(first, count, tensor):
    shares.append((first, count))
    tensor += torch.from_numpy(fn(first, count).astype(np.int64))
on_dev = L.sim.simulate_until(None, 20, min_frame_errors=12, max_frames=2000, batch=150, first_frame=40, rank=rank,
                              world=world, device="cpu", accumulate_fn=acc)
assert on_dev.dtype == np.uint64 and (on_dev == alone).all(), (on_dev[:3], alone[:3])
assert len(shares) == done // 150
print("STOP", int(on_dev[0]), int(on_dev[1]))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run_gloo_workers(script, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
        outs.append(out)
    return outs


def test_world_size_two_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    _run_gloo_workers(script, 2)


def test_stop_batch_does_not_depend_on_the_number_of_ranks(tmp_path):
    """The stop rule of an error-rate point (>= N frame errors, looked at once per batch of global frames) with 1, 2
    and 3 ranks on gloo: every rank of every world size stops after the same batch with the same counters, on the
    host-counter path and on the tensor path alike (the worker also checks both against an unsharded run)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    stops = set()
    for world in (1, 2, 3):
        for out in _run_gloo_workers(script, world):
            stops.add([l for l in out.splitlines() if l.startswith("STOP")][0])
    assert len(stops) == 1, stops


def test_bench_plumbing_world_size_two_gloo():
    """bench.py's own stepping, counter all-reduce, max-over-ranks timing and JSON line with two ranks on gloo
    and a stand-in decoder (--plumbing-test), so that the first real multi-GPU run cannot fail on plumbing."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    F, steps = 1000, 2
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps),
                                       "--warmup", "1", "--frames", str(F), "--plumbing-test"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err
        outs.append(out)
    assert outs[1].strip() == ""                      # only rank 0 prints
    lines = [l for l in outs[0].splitlines() if l.strip()]
    assert len(lines) == 1                            # exactly one line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak" and d["unit"] == "Gbit/s"
    # whole-job value = frames of both ranks x k / the slowest rank's time
    k = d["config"]["n"] - d["config"]["m"]
    assert abs(d["value"] - 2 * F * steps * k / (d["ms_per_step"] * steps / 1e3) / 1e9) < 1e-9 * max(1.0, d["value"])
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["allreduce_ms_per_step"]) == 2
    assert d["ms_per_step"] >= max(pr["ms_per_step"]) - 1e-6       # the maximum over the ranks
    # rank 1's stand-in decode sleeps twice as long: rank 0 waits for it inside the all-reduce
    assert pr["own_work_ms_per_step"][1] > pr["own_work_ms_per_step"][0]
    assert pr["allreduce_ms_per_step"][0] > pr["allreduce_ms_per_step"][1]
    # counters of the last step, summed over both ranks: global frames 0 .. 2F-1, every 7th a frame error
    assert d["counters"]["frames"] == 2 * F and d["counters"]["frame_errors"] == len(range(0, 2 * F, 7))


def test_rates_from_counters(pkg):
    c = np.zeros(16 + 5, np.uint64)
    c[0], c[1], c[2], c[4], c[8], c[9], c[10] = 100, 4, 12, 3, 300, 20, 50
    r = pkg.sim.rates(c, m=8, n=24)
    assert r["FER"] == 0.04 and r["IFER"] == 0.03 and r["IBER"] == 12 / (16 * 100)
    assert r["avg_iters_correct"] == 300 / 96 and r["avg_iters_wrong"] == 5.0 and r["channel_BER"] == 50 / 1600


def _build_host_example(out):
    pkg_dir = os.path.join(ROOT, "ldpc-simulator_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(pkg_dir, "host", "example_decode.cpp"),
           "-L", pkg_dir, "-lldpc_amd", "-Wl,-rpath," + pkg_dir, "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_cpp_host_mirror_builds_against_the_library(tmp_path, pkg):
    """host/LogDecoder.hpp (the C++ mirror of LogDecoder.decode and its DecodeResult record,
    LogDecoder.java:15-17) compiles warning-free and links against the C ABI."""
    pkg.capi.load_library()
    _build_host_example(tmp_path / "example_decode")
    assert (tmp_path / "example_decode").exists()


@pytest.mark.gpu
def test_cpp_host_mirror_runs(tmp_path):
    """The known answer H = [1 1 1], r = 000, e = 0.1 through the C++ mirror: 000, 1 iteration, syndrome 0."""
    _build_host_example(tmp_path / "example_decode")
    r = subprocess.run([str(tmp_path / "example_decode")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "decodedCode = 000  iterationNum = 1  syndrome = 0" in r.stdout, r.stdout + r.stderr


def test_dvbs2_shaped_builder(pkg):
    """Degree profile of the long DVB-S2 rate-1/2 code: 12960 columns of degree 8, 19440 of degree 3, a parity
    staircase of degree 2 whose last column has degree 1; 226,799 edges; every check but the first of degree 7."""
    import numpy as np
    m, n, rows, cols = pkg.codes.dvbs2_like(64800, seed=3)
    assert (m, n, len(rows)) == (32400, 64800, 226799)
    dc, dv = pkg.codes.degrees(m, n, rows, cols)
    assert (dv[:12960] == 8).all() and (dv[12960:32400] == 3).all() and (dv[32400:-1] == 2).all() and dv[-1] == 1
    assert dc[0] == 6 and (dc[1:] == 7).all()
    key = rows.astype(np.int64) * n + cols
    assert (np.diff(key) > 0).all()                      # row-major order, no repeated entry
    # the parity part is the staircase: check i holds columns k + i and k + i - 1
    k = n - m
    par = cols >= k
    assert (np.abs((cols[par] - k) - rows[par]) <= 1).all() and ((cols[par] - k) <= rows[par]).all()
