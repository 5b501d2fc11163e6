"""`python bench.py --gpus N` launches its own N ranks (no torchrun around it): the re-exec logic on the CPU, and end to end -- the
ranks start, find no GPU here and say so (the product path has no CPU fallback).  The reference needs no external launcher either
(src/sampling.jl:266,292: `pmap` over workers, one GPU each: src/util_parallel.jl:73-102)."""
import argparse
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_self_launch_command():
    a = argparse.Namespace(gpus=1)
    assert bench.self_launch_command(a, ["--gpus", "1"], {}) is None                         # N = 1: this process is the job
    a = argparse.Namespace(gpus=4)
    assert bench.self_launch_command(a, ["--gpus", "4"], {"WORLD_SIZE": "4", "RANK": "0"}) is None     # already a rank
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    cmd = bench.self_launch_command(a, argv, {})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                                               # the user's flags reach every rank unchanged


def test_self_launch_shares_the_host_cores_between_the_ranks():
    """eight ranks on one host: each gets an eighth of the cores for its NumPy / pocketfft / BLAS set-up work, unless the caller chose"""
    env = bench.launch_env({"PATH": "/bin"}, 8, ncpu=256)
    assert env["OMP_NUM_THREADS"] == env["CMBL_ORACLE_FFT_WORKERS"] == env["MKL_NUM_THREADS"] == "32"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    env = bench.launch_env({"OMP_NUM_THREADS": "4"}, 8, ncpu=4)
    assert env["OMP_NUM_THREADS"] == "4" and env["CMBL_ORACLE_FFT_WORKERS"] == "1"


def test_one_gpu_per_rank():
    assert bench.assign_devices(8, 8, "nccl") == list(range(8))
    assert bench.assign_devices(2, 8, "nccl") == [0, 1]
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.assign_devices(2, 1, "nccl")                                                   # fewer GPUs than ranks: loud, not shared
    assert bench.assign_devices(4, 1, "gloo") == [0, 0, 0, 0]                                # the test aid may share
    with pytest.raises(SystemExit, match="needs a GPU"):
        bench.assign_devices(1, 0, "nccl")


def test_clock_ramp_stops_when_blocks_agree():
    import time
    dur = iter([0.004, 0.004, 0.003, 0.002] + [0.001] * 100)
    cur = [None]
    n_steps = [0]
    def step():
        if n_steps[0] % 5 == 0:
            cur[0] = next(dur)
        n_steps[0] += 1
        time.sleep(cur[0])
    n, ms = bench.clock_ramp(step, lambda: None, tol=0.2)
    assert n == n_steps[0] and 10 <= n <= 40 and n % 5 == 0


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="CPU-only check: with a GPU the ranks would run the benchmark")
def test_plain_python_bench_gpus_2_starts_two_ranks():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0", "--nside", "64"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-3000:]                     # both ranks ran main() and failed loudly


def test_dry_run_of_the_eight_rank_launch():
    """`python bench.py --dry-run-ranks 8`: the whole N-rank path except the kernels -- self-launch, eight processes, one (mocked) device
    each, seeds base + 1000 rank, barriers, MAX over ranks, the gather and the collective report -- runs on the CPU over gloo."""
    import json
    r = subprocess.run([sys.executable, "bench.py", "--dry-run-ranks", "8", "--steps", "5", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                                         # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["steps"] == 5 and out["scaling"] == "weak"
    c = out["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 8
    assert [m["rank"] for m in c["ranks"]] == list(range(8))
    assert sorted(m["device"] for m in c["ranks"]) == list(range(8))                         # one device per rank, each used once
    assert len({m["pid"] for m in c["ranks"]}) == 8                                          # eight processes
    assert [m["seeds"] for m in c["ranks"]] == [[1 + 1000 * k, 2 + 1000 * k, 3 + 1000 * k] for k in range(8)]
    assert out["logpdf"] == [1.0 + 1000 * k for k in range(8)]                               # the per-chain gather, in rank order
    assert out["value"] > 0 and out["ms_per_step"] >= 1.0                                    # MAX over ranks of a >= 1 ms stand-in step


def test_dry_run_refuses_fewer_devices_than_ranks():
    # under a launcher with more ranks than (mocked) devices the one-GPU-per-rank rule still fires
    env = dict(os.environ, WORLD_SIZE="4", RANK="3", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, "bench.py", "--dry-run-ranks", "2", "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "only 2 GPU" in r.stderr, r.stderr[-2000:]
