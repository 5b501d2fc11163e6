"""N > 1 ranks with the kernels in the loop, on ONE GPU (gloo backend, every rank uses device 0): the multi-GPU logic of the
drivers and of bench.py is exercised end to end; on an 8-GPU node the same code runs with backend nccl (= RCCL over xGMI), one
rank per GPU.  BASELINE config 4 is the shape: 8 independent T+QU sample_joint chains, one per rank, results gathered, one chain file."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def torch_device_count():
    import torch
    return torch.cuda.device_count()


def _run(nproc, script_args, port, env=None, timeout=1200):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {})), capture_output=True, text=True, timeout=timeout)


def test_sample_joint_8_ranks_IQU_one_chain_file():
    r = _run(8, ["tools/gpu_dist_check.py"], 29541, env=dict(CMBL_DIST_POL="IP", CMBL_DIST_NCH="8", CMBL_DIST_SKIP_MARG="1"))
    assert r.returncode == 0 and "DIST_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_map_marg_and_chains_2_ranks():
    r = _run(2, ["tools/gpu_dist_check.py"], 29542)
    assert r.returncode == 0 and "DIST_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_two_ranks_prints_one_line_consistent_with_one_rank():
    """PLAIN `python bench.py --gpus 2` (no torchrun around it -- the form the driver uses; bench.py starts its own two ranks): ONE JSON
    line from rank 0, n_gpus = 2, value = 2 chains' worth of steps over the max-over-ranks time; the --gpus 1 line has the same keys
    (SCALE and BENCH records agree in form).  The same command under torch.distributed.run gives the same line."""
    common = ["--steps", "5", "--warmup", "1", "--nside", "256", "--no-cpu-baseline", "--no-roofline"]
    r2 = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dist-backend", "gloo"] + common, cwd=ROOT, capture_output=True, text=True,
                        timeout=1200, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    r2t = _run(2, ["bench.py", "--gpus", "2", "--dist-backend", "gloo"] + common, 29543)
    lt = [l for l in r2t.stdout.splitlines() if l.startswith("{")]
    assert r2t.returncode == 0 and len(lt) == 1 and json.loads(lt[0])["n_gpus"] == 2, r2t.stdout[-2000:] + r2t.stderr[-2000:]
    lines = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert r2.returncode == 0 and len(lines) == 1, r2.stdout[-2000:] + r2.stderr[-2000:]
    d2 = json.loads(lines[0])
    r1 = subprocess.run([sys.executable, "bench.py"] + common, cwd=ROOT, capture_output=True, text=True, timeout=600)
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    # the N > 1 line additionally reports who took part in the collective (backend, world size, device of every rank)
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1 and set(d1) == set(d2) - {"collective"} and len(d2["logpdf"]) == 2
    assert d2["collective"]["world_size"] == 2 and [r["rank"] for r in d2["collective"]["ranks"]] == [0, 1]
    # without the gloo test aid two ranks on a one-GPU box are refused loudly (one rank per GPU, src/util_parallel.jl:73-102)
    if torch_device_count() == 1:
        bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert bad.returncode != 0 and "only 1 GPU" in bad.stderr, bad.stderr[-2000:]
    assert d2["scaling"] == "weak" and d2["metric"] == d1["metric"] and d2["config"]["nside"] == 256
    assert abs(d2["value"] - 2 * 5 / (d2["ms_per_step"] * 5e-3)) < 1e-6 * d2["value"]


def test_nccl_backend_world_size_one():
    """The RCCL branch itself (`nccl` backend, device-resident payloads) with the only world size a one-GPU box offers:
    bench.py's init / barrier / all_reduce(MAX) / all_gather / all_gather_object path and chains.allreduce_sum /
    gather_chain_values on device tensors all execute; the JSON line reports the collective as RCCL sees it."""
    common = ["--steps", "3", "--warmup", "1", "--nside", "256", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
    r = _run(1, ["bench.py", "--gpus", "1", "--dist-backend", "nccl"] + common, 29544, env=dict(CMBL_BENCH_FORCE_DIST="1"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["collective"]["backend"] == "nccl" and d["collective"]["world_size"] == 1
    assert d["collective"]["ranks"][0]["rank"] == 0 and d["collective"]["ranks"][0]["device"] == 0
    code = ("import os, torch, torch.distributed as dist, numpy as np, cmblensing_jl_amd as C\n"
            "torch.cuda.set_device(0)\n"
            "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
            "t = torch.arange(6, dtype=torch.float64, device='cuda').reshape(3, 2)\n"
            "c = torch.view_as_complex(t.clone())\n"
            "buf = torch.view_as_real(c).contiguous(); dist.all_reduce(buf)\n"          # what allreduce_sum does on the nccl branch for world > 1
            "assert torch.equal(buf, t)\n"
            "assert C.allreduce_sum(c, dist) is c\n"
            "g = C.gather_chain_values([0, 1], np.array([[1., 2.], [3., 4.]]), 2, dist, 'cuda')\n"
            "assert g.tolist() == [[1., 2.], [3., 4.]]\n"
            "v = [torch.empty_like(t)]; dist.all_gather(v, t); assert torch.equal(v[0], t)\n"
            "dist.destroy_process_group(); print('NCCL_WS1_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29545"))
    assert r.returncode == 0 and "NCCL_WS1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
