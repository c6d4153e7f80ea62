#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): sliding windows decoded per second, [[144,12,12]] BB code,
circuit-level noise p = 0.003, (W,F) = (3,1) over 12 rounds -> 11 windows per shot, BP+OSD on the
shortened window matrix (osd_window semantics), batch = 4096 shots per GPU.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

With N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`, rendezvous on 127.0.0.1) and only relays
rank 0's JSON line; it refuses to run when fewer than N GPUs are visible -- it never falls back to one GPU.  Started
by torch.distributed.run (the driver's way) it is one rank of the job.

One "step" = one pass of the whole hot path over one batch of synthetic shots: a single launch of the
sliding-window pipeline kernel that decodes shots x 11 windows with commit and residual-syndrome update.
Detector data is sampled from the DEM on the device BEFORE the timed region and is resident in HBM when timing
starts.  Shots are sharded over ranks with no data-path collective (weak: 4096 shots per GPU; strong: a fixed
total split contiguously); one RCCL all_gather of the per-shot decisions (observable flips + flagged bit,
slidingwindowdecoder_amd.distributed.gather_decisions) closes the job inside the timed region.

Rank 0 prints ONE JSON line.  `roofline`: the kernel keeps its messages in LDS, so HBM is not what bounds it;
`frac` is the largest of the measured utilisations (the CU's LDS instruction path, LDS array, VALU issue, HBM), each = busy time at the 2.4 GHz
peak clock from the committed rocprofv3 counters (profiles/) / the kernel time measured live with HIP events --
at most 1 by construction.  SURVEY 8(d)'s algorithmic-bytes figure is reported next to it as `achieved_algorithmic`
(it exceeds the HBM peak: that is the traffic the LDS-resident design avoids, not a utilisation).
`cpu_baseline`: the CPU oracle (bit-exact port of the reference's Cython path) timed on this box's host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DECODER_KW = dict(pre_max_iter=8, post_max_iter=200, ms_scaling_factor=1.0, new_n=None, osd_method="osd_cs")
GDG_KW = dict(decoder="bpgdg_decoder", max_iter=8, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
              max_tree_branch_step=10, max_side_branch_step=10)  # `Sliding Window GDG.ipynb` cell 3
# --workload: the headline (BASELINE configs[1], what the driver measures) or one of the other circuit-level configurations
# under the same launcher / sharding / gather (north star: "reported at 1, 2, 4 and 8 GPUs")
WORKLOADS = {
    "headline": dict(problem=dict(), metric="sliding windows decoded/s, [[144,12,12]] BB p=0.003",
                     desc="configs[1]: [[144,12,12]] BB, circuit-level p=0.003, 12 rounds, (W,F)=(3,1) -> 11 windows/shot, "
                          "osd_window(pre=8, post=200, alpha=1.0, osd_cs order %d)"),
    "gdg": dict(problem=dict(), metric="sliding windows decoded/s, [[144,12,12]] BB p=0.003, bpgdg_decoder",
                desc="configs[2]: [[144,12,12]] BB, circuit-level p=0.003, (3,1) windows, bpgdg_decoder(max_iter=8, T=6, R=25, D=3, S=10)"),
    "bb288": dict(problem=dict(N=288, W=4, F=1), metric="sliding windows decoded/s, [[288,12,18]] BB p=0.003",
                  desc="configs[3]: [[288,12,18]] BB, circuit-level p=0.003, 12 rounds, (W,F)=(4,1), "
                       "osd_window(pre=8, post=200, alpha=1.0, osd_cs order %d)"),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PEAK_CLOCK_HZ = 2.4e9   # same guide: max clock
NUM_CU, SIMD_PER_CU = 256, 4
PROFILE_TAG = "r02"
STUB = os.environ.get("SWD_BENCH_STUB") == "1"  # launcher test on CPU: gloo + a stand-in decoder, never a measurement


def build_problem(N=144, p=0.003, rounds=12, W=3, F=1):
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.windows import plan_windows
    code, A, B = bb_code(N)
    dem = bb_dem(code, A, B, p, rounds)
    return plan_windows(dem.chk, dem.obs, dem.priors, N // 2, W, F, method=1)


def algorithmic_bytes(plan, stats, pre_max_iter):
    """per BP iteration on a graph with E live edges, n live VNs, m live CNs and
    fp64 messages  B_iter = 40 E + 17 n + 2 m ; per window: sum over the iterations actually executed
    (full graph for the pre phase, shortened graph for the post phase) + sort 16 n per sort
    + OSD 2 * ceil((new_n+1)/64) * 8 bytes per GF(2) row addition applied + I/O (m + n).
    stats: int array [shots, W, 8] written by the kernel."""
    total = 0.0
    for wi, w in enumerate(plan.windows):
        m, n = w.mat.shape
        E = w.mat.nnz
        new_n = min(n, 2 * m)
        st = stats[:, wi, :].astype(np.float64)
        cls = (stats[:, wi, 0] & 0xFF)
        pre_it, post_it = st[:, 2], st[:, 3]
        full = 40.0 * E + 17.0 * n + 2.0 * m
        short = 40.0 * st[:, 6] + 17.0 * st[:, 4] + 2.0 * st[:, 5]
        b = pre_it * full + post_it * short
        b += (cls >= 1) * 16.0 * n            # history sort before shortening
        b += (cls == 2) * 16.0 * n            # OSD ordering
        b += st[:, 7] * 2.0 * ((new_n + 1 + 63) // 64) * 8.0
        b += m + n
        total += b.sum()
    return total


def irreducible_hbm_bytes(plan, shots):
    """What one launch has to move through HBM whatever the kernel does: detector bytes in, committed faults out,
    the per-window statistics and per-shot decisions out, and the residual syndrome + accumulator record handed from
    each window of a shot to the next (written once, read once; windows of a shot run on different CUs)."""
    num_det, num_col = plan.chk.shape
    W = len(plan.windows)
    state = 16 + (num_det + 15) // 16 * 16
    return shots * (num_det + num_col + W * 8 * 4 + 8 + (W - 1) * 2 * state)


def cpu_baseline_worker(args):
    """Runs in a spawned process BEFORE any GPU initialisation: the oracle (oracle/swd_oracle.c) over
    the sliding-window loop for `shots` shots; returns (windows, seconds)."""
    seed, shots, order = args
    from oracle import oracle as O
    from slidingwindowdecoder_amd.windows import sample_dem
    import scipy.sparse as sp
    plan = build_problem()
    det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=seed)
    kw = dict(DECODER_KW, osd_order=order)
    decs = [O.osd_window(w.mat, channel_probs=w.prior, **kw) for w in plan.windows]
    chk_t = sp.csr_matrix(plan.chk.T.astype(np.int32))
    total = np.zeros((shots, plan.chk.shape[1]), np.uint8)
    cur = det.copy()
    t0 = time.perf_counter()
    for w, d in zip(plan.windows, decs):
        out, _ = d.decode_batch(cur[:, w.row0:w.row1])
        total[:, w.col0:w.col0 + w.commit] = out[:, :w.commit]
        cur = ((det + (sp.csr_matrix(total) @ chk_t).toarray()) % 2).astype(np.uint8)
    return shots * len(plan.windows), time.perf_counter() - t0


def cpu_baseline(order, shots_per_core=192):
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 32))
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, [(1000 + i, shots_per_core, order) for i in range(cores)])
    wall = time.perf_counter() - t0
    windows = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": windows / busy, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{shots_per_core} shots x 11 windows per core on {cores} processes (oracle/swd_oracle.c, "
                      f"bit-exact port of the reference's Cython osd_window; decode loop only, {busy:.1f} s; "
                      f"{wall:.1f} s incl. setup)",
            "per_core": windows / busy / cores}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shots", type=int, default=4096, help="shots per GPU per step (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline",
                    help="headline = BASELINE configs[1] (default; the only one with roofline / cpu_baseline); gdg, bb288 = configs[2], [3]")
    ap.add_argument("--total-shots", type=int, default=4096 * 8, help="shots per step over all GPUs (strong scaling)")
    ap.add_argument("--osd-order", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-order10", action="store_true", help="skip the extra osd_cs order-10 measurement (N = 1)")
    ap.add_argument("--distinct-batches", type=int, default=4, help="pre-sampled batches cycled over the steps")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 without a torch.distributed.run environment: start the N ranks as child processes."""
    if not STUB:
        import torch  # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to measure fewer GPUs than asked")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


class StubEngine:
    """Stand-in for the device pipeline in the launcher test (SWD_BENCH_STUB=1): decisions are a pure function of
    the global shot number, so the gathered result can be checked; nothing it reports is a measurement."""

    def __init__(self, args, rank, lo, hi):
        import torch
        self.torch, self.lo, self.hi = torch, lo, hi
        self.shot = torch.zeros((hi - lo, 2), dtype=torch.int32)
        self.W = 11

    @staticmethod
    def expected(lo, hi):
        g = np.arange(lo, hi, dtype=np.int64)
        return np.stack([(g * 2654435761) % 4093, g % 2], axis=1).astype(np.int32)

    def step(self, i):
        self.shot.copy_(self.torch.from_numpy(self.expected(self.lo, self.hi)))

    def sync(self):
        pass


class GpuEngine:
    def __init__(self, args, rank, local_rank, lo, hi, plan, order, workload="headline"):
        import torch
        from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
        self.torch = torch
        self.dev = torch.device("cuda", local_rank)
        self.plan, self.W = plan, len(plan.windows)
        kw = dict(GDG_KW) if workload == "gdg" else dict(DECODER_KW, osd_order=order)
        self.dec = SlidingWindowDecoder(plan, device=local_rank, **kw)
        shots = hi - lo
        self.nb = max(1, min(args.distinct_batches, args.steps + args.warmup))
        # synthetic shots sampled from the DEM on the device: Philox stream keyed by the global shot number, so the
        # data of a shot does not depend on the number of ranks
        sampler = DemSampler(plan.chk, plan.obs, plan.priors, device=local_rank)
        self.dets, self.obs_true = [], []
        for i in range(self.nb):
            det, flips = sampler.sample_device(shots, seed=20240318, first_shot=i * (1 << 24) + lo)
            self.dets.append(det)
            self.obs_true.append(flips.cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
        self.total = torch.empty((shots, plan.chk.shape[1]), dtype=torch.uint8, device=self.dev)
        self.stats = torch.empty((shots, self.W, 8), dtype=torch.int32, device=self.dev)
        self.shot = torch.empty((shots, 2), dtype=torch.int32, device=self.dev)

    def step(self, i):
        self.dec.decode_device(self.dets[i % self.nb], total=self.total, stats=self.stats, min_pm=None, shot_result=self.shot)

    def sync(self):
        self.torch.cuda.synchronize()


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def roofline(alg_bytes, avg_kernel_s, shots, plan):
    """Utilisation of the three resources the kernel could be bound by, from the committed per-launch counters
    (separate rocprofv3 --pmc passes of this same command, profiles/) and the live kernel time."""
    sq = load_profile(f"{PROFILE_TAG}_sq_counters.json") or load_profile("r01_sq_counters.json")
    sq_src = f"profiles/{PROFILE_TAG}_sq_counters.json" if load_profile(f"{PROFILE_TAG}_sq_counters.json") else "profiles/r01_sq_counters.json"
    hb = load_profile("hbm_traffic.json")
    traffic = hb.get("hbm_bytes_per_launch") if hb else None
    fr = {}
    if sq:
        c = sq["per_launch_mean"]
        if "SQ_ACTIVE_INST_VALU" in c:  # quad-cycles of VALU issue summed over all waves -> busy seconds per SIMD at peak clock
            busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (NUM_CU * SIMD_PER_CU) / PEAK_CLOCK_HZ
            fr["valu_issue"] = {"frac": busy / avg_kernel_s, "busy_ms_at_peak_clock": busy * 1e3,
                                "counter": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x 2.4 GHz)", "source": sq_src}
        if "SQ_ACTIVE_INST_LDS" in c:   # quad-cycles a wave spends issuing LDS instructions, summed over all waves: the LDS unit is one per CU
            busy = c["SQ_ACTIVE_INST_LDS"] * 4.0 / NUM_CU / PEAK_CLOCK_HZ
            fr["lds_pipe"] = {"frac": busy / avg_kernel_s, "busy_ms_at_peak_clock": busy * 1e3,
                              "counter": "SQ_ACTIVE_INST_LDS x 4 / (256 CUs x 2.4 GHz): time the waves of a CU spend issuing LDS instructions "
                                         "(a ds_write_b64 occupies the CU's LDS path ~6 cycles, a ds_read_b64 ~2: scripts/ubench/issue_rates.hip)",
                              "source": sq_src}
        if "SQ_LDS_IDX_ACTIVE" in c:    # LDS-array cycles summed over the CUs
            busy = c["SQ_LDS_IDX_ACTIVE"] / NUM_CU / PEAK_CLOCK_HZ
            fr["lds"] = {"frac": busy / avg_kernel_s, "busy_ms_at_peak_clock": busy * 1e3,
                         "bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"],
                         "counter": "SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz)", "source": sq_src}
    if traffic:
        fr["hbm_measured"] = {"frac": traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS, "GBps": traffic / avg_kernel_s / 1e9,
                              "counter": "(2 x FETCH_SIZE + WRITE_SIZE) per launch / 8 TB/s", "source": hb.get("source")}
    irr = irreducible_hbm_bytes(plan, shots)
    out = {"kernel": "swd::pipeline_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
           "traffic": traffic, "irreducible_hbm_bytes": irr,
           "traffic_over_irreducible": (traffic / irr) if traffic else None,
           "algorithmic_bytes_per_launch": alg_bytes,
           "achieved_algorithmic": alg_bytes / avg_kernel_s / 1e9, "achieved_algorithmic_unit": "GB/s",
           "achieved_algorithmic_over_hbm_peak": alg_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBS,
           "fractions": fr,
           "note": "messages never leave LDS, so SURVEY 8(d)'s algorithmic bytes (40E+17n+2m per executed BP iteration + "
                   "sort + OSD row adds + I/O) exceed what HBM could carry; frac = the highest measured utilisation among "
                   "the CU's LDS instruction path, the LDS array, VALU issue and HBM (counters from profiles/, time measured "
                   "here with HIP events).  The launch scales 1.76x from one to two workgroups per CU and not at all from two "
                   "to three (DESIGN.md section 4)"}
    if sq and "SQ_ACTIVE_INST_ANY" in sq["per_launch_mean"] and sq["per_launch_mean"].get("SQ_WAVE_CYCLES"):
        c = sq["per_launch_mean"]  # how the waves spend their cycles (not a capacity: two waves of a SIMD can issue different kinds together)
        out["wave_cycles"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "parked_waitcnt_or_barrier": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                              "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]}
    out = dict(out)
    if fr:
        bound = max(fr, key=lambda k: fr[k]["frac"])
        out.update({"bound": {"lds_pipe": "lds", "valu_issue": "valu", "lds": "lds", "hbm_measured": "hbm"}[bound], "frac": fr[bound]["frac"]})
        if bound == "hbm_measured":
            out.update({"achieved": fr[bound]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s"})
        else:
            out.update({"achieved": fr[bound]["busy_ms_at_peak_clock"], "peak": avg_kernel_s * 1e3,
                        "unit": "ms busy at 2.4 GHz per launch (of the launch's duration)"})
    else:
        out.update({"bound": "hbm", "frac": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s"})
    return out


def time_steps(engine, args, dist, world, total_shots):
    """W untimed steps, then exactly K timed steps + the gather of the decisions, bracketed by barriers and
    device synchronisation; returns (elapsed seconds = max over ranks, gathered decisions)."""
    from slidingwindowdecoder_amd.distributed import gather_decisions
    import torch
    for i in range(args.warmup):
        engine.step(i)
    engine.sync()
    if world > 1:
        dist.barrier()
    engine.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        engine.step(args.warmup + i)
    gathered = gather_decisions(engine.shot, total_shots)  # per-shot decisions of the last step, over RCCL
    engine.sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=engine.shot.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, gathered


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the job has WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not STUB and args.workload == "headline":
        cpu = cpu_baseline(args.osd_order)  # before the GPU is touched (spawned workers)

    import torch
    import torch.distributed as dist
    from slidingwindowdecoder_amd.distributed import shard_bounds
    if not STUB:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus

    total_shots = args.shots * world if args.scaling == "weak" else args.total_shots
    lo, hi = shard_bounds(total_shots, rank, world)
    wl = WORKLOADS[args.workload]
    headline = args.workload == "headline"
    plan = None if STUB else build_problem(**wl["problem"])
    engine = StubEngine(args, rank, lo, hi) if STUB else GpuEngine(args, rank, local_rank, lo, hi, plan, args.osd_order, args.workload)
    W = engine.W
    if not STUB:
        engine.dec.set_timing(True)  # HIP events around every kernel launch, on the launch stream
    elapsed, gathered = time_steps(engine, args, dist, world, total_shots)
    assert gathered.shape[0] == total_shots, (gathered.shape, total_shots)

    line = {
        "metric": wl["metric"],
        "value": total_shots * W * args.steps / elapsed,
        "unit": "windows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "timed_region_s": elapsed,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "stub (launcher test, not a measurement)" if STUB else "synthetic",
    }
    if STUB:
        ok = np.array_equal(gathered.numpy(), StubEngine.expected(0, total_shots))
        line["config"] = {"workload": "launcher test", "world_size": world, "shots_total": total_shots,
                          "shots_this_rank": hi - lo, "gather_ok": bool(ok)}
        if rank == 0:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        if not ok:
            raise SystemExit("gathered decisions differ from the expected ones")
        return

    kern_ms, launches = engine.dec.get_timing()
    engine.dec.set_timing(False)
    engine.dec.check_status()  # raises if any window of any launch gave up waiting for its predecessor

    # accounting (outside the timed region), on this rank's shard of the last step
    st = engine.stats.cpu().numpy()
    last = (args.warmup + args.steps - 1) % engine.nb
    sr = engine.shot.cpu().numpy()
    logical = (sr[:, 0].astype(np.int64) != engine.obs_true[last]) | (sr[:, 1] != 0)
    alg_bytes = algorithmic_bytes(plan, st, DECODER_KW["pre_max_iter"]) if headline else None
    cls = np.bincount((st[..., 0] & 0xFF).ravel(), minlength=7)
    avg_kernel_s = kern_ms / max(launches, 1) / 1e3

    order10 = None
    if rank == 0 and world == 1 and headline and not args.no_order10 and args.osd_order != 10:
        # the notebooks' default OSD-CS order 10 on the same batches (a second, untimed-by-the-driver loop)
        e10 = GpuEngine(args, rank, local_rank, lo, hi, plan, 10)
        k10 = max(1, min(args.steps, 5))
        e10.step(0); e10.sync()
        t0 = time.perf_counter()
        for i in range(k10):
            e10.step(i)
        e10.sync()
        order10 = total_shots * W * k10 / (time.perf_counter() - t0)
        e10.dec.check_status()

    if rank == 0:
        line["config"] = {
            "workload": (wl["desc"] % args.osd_order if "%d" in wl["desc"] else wl["desc"]) + ", "
                        + (f"{args.shots} shots per GPU per step" if args.scaling == "weak" else f"{total_shots} shots per step split over the GPUs"),
            "world_size": world, "shots_total": total_shots, "shots_rank0": hi - lo, "windows_per_shot": W,
            "parallelism": f"shots sharded over {world} GPU(s), no data-path collective; one all_gather of 8 B per shot",
            "exit_classes_pre_post_osd_rank0": [int(cls[0]), int(cls[1]), int(cls[2])],
            "sched_faults": int(cls[6]),
            "logical_errors_last_step_rank0": int(logical.sum()),
            "osd_cs_order10_windows_per_s": order10,
        }
        # roofline and CPU baseline belong to the headline kernel and its committed counter profile
        line["roofline"] = roofline(alg_bytes, avg_kernel_s, hi - lo, plan) if headline else None
        line["cpu_baseline"] = cpu if headline else None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
