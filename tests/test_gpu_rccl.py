"""RCCL readiness on the one GPU this build can reach: the bench started the driver's way -- `python -m torch.distributed.run
--nproc-per-node 1 ... bench.py --gpus 1` -- initialises backend nccl (= RCCL) with device_id, keeps libswd_hip.so on torch's
preloaded HIP runtime (_lib.py), closes the timed region with the all_gather of the decisions on CUDA tensors, and measures the
same rate as the plain one-process run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "1", "--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--no-order0", "--no-other-workloads"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SWD_BENCH_STUB")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(lines[0])


def test_one_rank_under_torchrun_on_rccl_equals_plain_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    bench = os.path.join(ROOT, "bench.py")
    plain = _line(subprocess.run([sys.executable, bench, *ARGS], capture_output=True, text=True, env=_env(), timeout=900))
    dist = _line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                                 "--master-port", str(port), bench, *ARGS], capture_output=True, text=True, env=_env(), timeout=900))
    assert plain["config"]["collective_ranks"] == 0 and plain["config"]["collective_backend"] is None
    assert dist["config"]["collective_backend"] == "nccl" and dist["config"]["collective_ranks"] == 1
    assert dist["n_gpus"] == 1 and dist["config"]["sched_faults"] == 0
    # same shots, same decisions
    assert dist["config"]["exit_classes_pre_post_osd_rank0"] == plain["config"]["exit_classes_pre_post_osd_rank0"]
    assert dist["config"]["logical_errors_last_step_rank0"] == plain["config"]["logical_errors_last_step_rank0"]
    ratio = dist["value"] / plain["value"]
    print(f"windows/s plain {plain['value']:.4g}, one rank on RCCL {dist['value']:.4g}, ratio {ratio:.4f}")
    assert abs(ratio - 1.0) < 0.06, (plain["value"], dist["value"])  # (two separate processes, 0.6 s of timed region each: 3-4 % apart from run to run)


def test_gather_decisions_on_a_cuda_tensor_in_a_one_rank_rccl_group():
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from slidingwindowdecoder_amd import _lib\n"
        "from slidingwindowdecoder_amd.distributed import gather_decisions\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "assert _lib.lib().swd_device_count() >= 1  # our library sees the device through torch's HIP runtime\n"
        "t = torch.arange(14, dtype=torch.int32, device='cuda').reshape(7, 2)\n"
        "g = gather_decisions(t, 7)\n"
        "assert g.is_cuda and torch.equal(g, t)\n"
        "dist.barrier(); dist.destroy_process_group(); print('ok')\n")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(_env(), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
