"""bench.py's own multi-process path on CPU: `python bench.py --gpus 2` without a torch.distributed.run environment
has to start two ranks itself, shard the shots, gather every rank's decisions and print ONE line with n_gpus = 2.
SWD_BENCH_STUB=1 swaps the device pipeline for a stand-in (gloo backend) -- the launcher, sharding, gather, timing
bracket and JSON contract are the code under test, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """a rendezvous port nobody listens on (concurrent test runs must not collide on a hard-coded one)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def run_bench(*argv, env_extra=None, timeout=300):
    env = dict(os.environ, SWD_BENCH_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_self_launch_two_ranks_weak():
    r, lines = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--shots", "50")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout  # only rank 0 prints
    j = lines[0]
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3 and j["warmup"] == 1
    assert j["config"]["world_size"] == 2 and j["config"]["shots_total"] == 100 and j["config"]["gather_ok"]
    assert j["value"] > 0 and abs(j["timed_region_s"] * 1e3 / 3 - j["ms_per_step"]) < 1e-6
    assert "stub" in j["data"]


def test_self_launch_strong_scaling_uneven_shards():
    r, lines = run_bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--scaling", "strong", "--total-shots", "101")
    assert r.returncode == 0, r.stderr[-2000:]
    j = lines[0]
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["config"]["shots_total"] == 101 and j["config"]["shots_this_rank"] == 51 and j["config"]["gather_ok"]


def test_self_launch_eight_ranks_weak_and_strong():
    """the shape of the driver's 8-GPU run (stub decoder, gloo): eight ranks started by bench.py itself, weak scaling with 4096-shot
    semantics scaled down, and strong scaling with shards that do not divide evenly; the line describes the collective itself"""
    r, lines = run_bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--shots", "16", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    j = lines[0]
    c = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and c["world_size"] == 8 and c["shots_total"] == 128 and c["gather_ok"]
    assert c["collective_backend"] == "gloo" and c["collective_ranks"] == 8
    assert c["shots_per_rank"] == [16] * 8
    # the line diagnoses imbalance by itself: every rank's step time and gather time
    rt = c["rank_times"]
    assert len(rt["ms_per_step_per_rank"]) == 8 and len(rt["gather_ms_per_rank"]) == 8
    assert 0 < rt["ms_per_step_min_over_ranks"] <= rt["ms_per_step_max_over_ranks"] <= j["ms_per_step"] + 1e-9
    assert rt["gather_ms_max_over_ranks"] == max(rt["gather_ms_per_rank"]) >= 0
    r, lines = run_bench("--gpus", "8", "--steps", "2", "--warmup", "0", "--scaling", "strong", "--total-shots", "203", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = lines[0]
    c = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and c["shots_total"] == 203 and c["gather_ok"]
    assert sum(c["shots_per_rank"]) == 203 and max(c["shots_per_rank"]) - min(c["shots_per_rank"]) <= 1
    assert c["shots_this_rank"] == c["shots_per_rank"][0]


def test_torchrun_launch_two_ranks():
    """the driver's own launch line for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N (stub, gloo)"""
    env = dict(os.environ, SWD_BENCH_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--shots", "21"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["config"]["collective_ranks"] == 2 and lines[0]["config"]["gather_ok"]


def test_single_process_default():
    r, lines = run_bench("--steps", "2", "--warmup", "1", "--shots", "10")
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines[0]["n_gpus"] == 1 and lines[0]["config"]["world_size"] == 1


def test_world_size_mismatch_is_an_error():
    r, _ = run_bench("--gpus", "4", "--steps", "1", "--warmup", "0", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_refuses_more_gpus_than_visible():
    """without the stub: this container has no GPU, so --gpus 2 must fail loudly instead of measuring one"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SWD_BENCH_STUB")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr


def test_roofline_is_a_bounded_utilisation():
    """`roofline.frac` is achieved / peak from ALGORITHMIC bytes -- lds_bytes_algorithmic / t / 79 TB/s for the LDS-resident kernels,
    SURVEY 8(d)'s bytes / t / 8 TB/s for the one whose messages live in HBM -- never a busy fraction (those grow with bank
    conflicts and padding and stay under `utilisation`, each below 1) and never the HBM-priced figure of an LDS-resident kernel;
    a kernel time that has moved away from the profiled one is flagged."""
    sys.path.insert(0, ROOT)
    import bench
    sq, sm, sq_src, sm_src = bench.find_profile("headline")
    assert sq and sm, "no committed counter profile of the headline kernel"
    t = sm["avg_ms"] * 1e-3
    r = bench.roofline("headline", "swd::pipeline_kernel", 1.49e11, 1.1e11, t, 1.2e8)
    assert r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == bench.LDS_MIX_PEAK_GBS == 79000.0
    assert r["frac"] == r["lds_bytes_algorithmic"] / t / 1e9 / bench.LDS_MIX_PEAK_GBS == r["lds_algorithmic_frac"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-15 and 0.0 < r["frac"] < 1.0
    u = r["utilisation"]
    assert {"lds", "valu", "hbm"} <= set(u) and all(0.0 < u[k]["frac"] < 1.0 for k in ("lds", "valu", "hbm"))
    assert u["max"] == max(u[k]["frac"] for k in ("lds", "valu", "hbm")) and u["busiest"] in ("lds", "valu", "hbm")
    assert r["frac"] < u["lds"]["frac"]  # the algorithm's own bytes are a part of what keeps the LDS pipeline busy
    assert "fractions" not in r
    assert "lds_pipe" in r["diagnostics"] and "lds_pipe" not in u  # wave time inside LDS instructions is no capacity
    assert r["achieved_algorithmic_over_hbm_peak"] > 1.0  # reported, but not as `frac`
    assert r["traffic"] >= r["irreducible_hbm_bytes"] > 0
    assert r["lds_bytes_moved"] > r["lds_bytes_algorithmic"] > 0 and r["lds_padding_factor"] > 1.0
    assert r["profile_stale"] is False and r["profile"]["counters"] == sq_src
    assert bench.roofline("headline", "k", 1.49e11, 1.1e11, t * 1.2, 1.2e8)["profile_stale"] is True
    # the streamed step: the same bytes over the step time
    r2 = bench.roofline("headline", "k", 1.49e11, 1.1e11, t, 1.2e8, step_s=t * 0.9, step_mode="two-lane stream")
    assert abs(r2["at_step_time"]["frac"] - r2["frac"] / 0.9) < 1e-12
    # HBM-resident messages (global144): SURVEY 8(d) bytes against the HBM peak
    g = bench.roofline("global144", "k", 2.0e10, None, 14.6e-3, None)
    assert g["bound"] == "hbm" and g["peak"] == 8000.0 and abs(g["frac"] - 2.0e10 / 14.6e-3 / 1e9 / 8000.0) < 1e-15


def test_cpu_baseline_carries_the_reference_figure():
    """the reference's own Cython osd_window, timed beside the port by tests/golden/time_reference.py (committed figure)"""
    sys.path.insert(0, ROOT)
    import bench
    f = bench.reference_cpu_figure()
    assert 300 < f["reference_cython_per_core"] < 1000 and 0.3 < f["port_vs_reference"] < 3.0
    assert "profiles/r06_cpu_reference_vs_port.json" in f["reference_figure_source"]


def test_one_rank_job_still_runs_the_collective():
    """started by torch.distributed.run with one rank, the bench initialises the process group and closes the timed
    region with the all_gather (gloo here; RCCL on the GPU box: tests/test_gpu_rccl.py)"""
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": free_port()}
    r, lines = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--shots", "23", env_extra=env)
    assert r.returncode == 0, r.stderr[-2000:]
    c = lines[0]["config"]
    assert c["collective_backend"] == "gloo" and c["collective_ranks"] == 1 and c["gather_ok"]
    r, lines = run_bench("--steps", "2", "--warmup", "1", "--shots", "23")  # plain run: no process group
    assert lines[0]["config"]["collective_ranks"] == 0


def test_other_workloads_share_the_launcher():
    """`--workload gdg|bb288|bp4` run configs[2] / configs[3] / the quaternary decoder under the same launcher, sharding and gather (stub: CPU)."""
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.WORKLOADS) == {"headline", "gdg", "gdg64", "bb288", "bp4", "global144"}
    assert bench.parse_args(["--workload", "bp4"]).shots == 65536 and bench.parse_args([]).shots == 4096
    assert bench.parse_args(["--workload", "bb288"]).workload == "bb288" and bench.parse_args([]).workload == "headline"
    assert bench.parse_args([]).osd_order == 10  # the notebooks' default is the headline
    r, lines = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--shots", "37", "--workload", "gdg")
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["gather_ok"] is True and "bpgdg_decoder" in lines[0]["metric"]
