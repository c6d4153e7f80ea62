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
Consecutive steps go through the product's streaming entry (include/swd.h: swd_pipeline_stream_push_dev -- two lanes, each with
its own HIP stream, launch slot and output buffers): step k + 1 is launched while step k still runs and its persistent grid takes
the workgroup slots that the tail of step k leaves empty.  Every one of the K timed steps starts and finishes inside the timed
region.  `--no-stream` times one launch at a time instead (also reported as config.single_stream_windows_per_s); the kernel
duration behind `roofline` is always measured one launch at a time, with HIP events on the launch stream, after the timed region.
Detector data is sampled from the DEM on the device BEFORE the timed region and is resident in HBM when timing
starts.  Shots are sharded over ranks with no data-path collective (weak: 4096 shots per GPU; strong: a fixed
total split contiguously); one RCCL all_gather of the per-shot decisions (observable flips + flagged bit,
slidingwindowdecoder_amd.distributed.gather_decisions) closes the job inside the timed region.

Rank 0 prints ONE JSON line.  `roofline.frac` = ALGORITHMIC bytes per launch / kernel time / peak of the memory that
binds the kernel -- achieved / peak, nothing that grows with padding or bank conflicts:
  LDS-resident kernels (headline, bb288, gdg, gdg64, bp4): the messages never leave LDS, so the binding memory is LDS;
        achieved = 32 B per live edge and executed BP iteration (each of the two passes reads and writes the edge's 8-byte message
        once; the counts come from the kernel's own statistics) / t; peak = 79 TB/s, what a half-read / half-write ds_*_b64 mix
        can move with all 256 CUs streaming (MI355X_MICROARCH.md, LDS: ~150 TB/s reads, 38-51 TB/s writes); bound = "lds"
  global144 (messages of the full graph in HBM): SURVEY 8(d)'s 40E+17n+2m bytes per executed iteration + sort + OSD + I/O / t
        against the 8 TB/s HBM peak; bound = "hbm"
`roofline.utilisation` keeps the BUSY fractions of rounds 3-5 (how occupied a unit is, padding and conflicts included -- not a
score): lds = (SQ_LDS_IDX_ACTIVE + 2 x SQ_INSTS_LDS_STORE) / (256 CUs x 2.4 GHz x t); valu = (4 x fp64-rate + 2 x other VALU
instructions) / (1024 SIMDs x 2.4 GHz x t); hbm = (2 x FETCH_SIZE + WRITE_SIZE) / (8 TB/s x t).  Counters: the committed
rocprofv3 profile of this same command (profiles/<tag>_<workload>_*); time: HIP events measured live; `profile_stale` says when
the live kernel time has moved away from the profiled one.  `traffic` = HBM bytes per launch from the PMC passes (gfx950 read-side
correction applied); `lds_bytes_moved` (SQ_INSTS_LDS x 512) / `lds_bytes_algorithmic` = the padding of the LDS traffic;
`achieved_algorithmic` = SURVEY 8(d)'s HBM-priced figure (exceeds the HBM peak for the LDS-resident kernels: the traffic the
design avoids).
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
for _k in ("pre_max_iter", "post_max_iter"):  # diagnostics only (scripts/lds_conflict_attribution.sh): attribute counters to a phase by differences
    if os.environ.get("SWD_BENCH_" + _k.upper()):
        DECODER_KW[_k] = int(os.environ["SWD_BENCH_" + _k.upper()])
GDG_KW = dict(decoder="bpgdg_decoder", max_iter=8, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
              max_tree_branch_step=10, max_side_branch_step=10)  # `Sliding Window GDG.ipynb` cell 3
# --workload: the headline (BASELINE configs[1], what the driver measures) or one of the other circuit-level configurations
# under the same launcher / sharding / gather (north star: "reported at 1, 2, 4 and 8 GPUs")
GDG64_KW = dict(GDG_KW, multi_thread=True, max_tree_depth=5, max_side_depth=6)  # BASELINE configs[2] as written: 64 hypotheses per shot
WORKLOADS = {
    "headline": dict(problem=dict(), metric="sliding windows decoded/s, [[144,12,12]] BB p=0.003",
                     desc="configs[1]: [[144,12,12]] BB, circuit-level p=0.003, 12 rounds, (W,F)=(3,1) -> 11 windows/shot, "
                          "osd_window(pre=8, post=200, alpha=1.0, osd_cs order %d)"),
    "gdg": dict(problem=dict(), metric="sliding windows decoded/s, [[144,12,12]] BB p=0.003, bpgdg_decoder",
                desc="configs[2]: [[144,12,12]] BB, circuit-level p=0.003, (3,1) windows, bpgdg_decoder(max_iter=8, T=6, R=25, D=3, S=10)"),
    "gdg64": dict(problem=dict(), metric="sliding windows decoded/s, [[144,12,12]] BB p=0.003, bpgdg_decoder(multi_thread=True), 64 hypotheses",
                  desc="configs[2] as written: [[144,12,12]] BB, circuit-level p=0.003, (3,1) windows, the reference's threaded ensemble "
                       "bpgdg_decoder(multi_thread=True, max_iter=8, T=6, R=25, D=5, S=6) = main + 31 tree threads x 2 leaves + 1 side thread"),
    "bb288": dict(problem=dict(N=288, W=4, F=1), metric="sliding windows decoded/s, [[288,12,18]] BB p=0.003",
                  desc="configs[3]: [[288,12,18]] BB, circuit-level p=0.003, 12 rounds, (W,F)=(4,1), "
                       "osd_window(pre=8, post=200, alpha=1.0, osd_cs order %d)"),
    # row J: osd_window on the UN-windowed detector error model as /root/reference/IBM.ipynb:119-135 configures it (936 x 8784, 30 672
    # edges = 245 KB of fp64 messages per shot: the large-graph kernels keep them in HBM, so here SURVEY 8(d)'s HBM roofline applies)
    "global144": dict(problem=dict(p=0.004, W=13, F=1), decoder_kw=dict(pre_max_iter=16, post_max_iter=1000),
                      metric="global decodes/s, [[144,12,12]] BB p=0.004, osd_window on the 936 x 8784 DEM", unit="decodes/s",
                      desc="IBM.ipynb decode(shorten=True): [[144,12,12]] BB, circuit-level p=0.004, 12 rounds, ONE window = the whole "
                           "936 x 8784 detector error model, osd_window(pre=16, post=1000, alpha=1.0, osd_cs order %d)"),
    # the quaternary decoder of configs[4], device-resident (the reference runs BP4 on code-capacity noise only: Misc.ipynb cell 2)
    "bp4": dict(problem=None, metric="bp4_osd decodes/s, [[144,12,12]] BB depolarizing p=0.02", unit="decodes/s",
                desc="bp4_osd(max_iter=100, alpha=0.625, osd_cs order 10) on [[144,12,12]] hx/hz, depolarizing code-capacity noise p=0.02 "
                     "(Misc.ipynb cell 2 setting), syndromes resident in HBM, persistent grid"),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PEAK_CLOCK_HZ = 2.4e9   # same guide: max clock
NUM_CU, SIMD_PER_CU = 256, 4
PROFILE_TAGS = ("r06", "r05", "r04", "r03")  # newest first; find_profile falls back to the round-2 headline files
STUB = os.environ.get("SWD_BENCH_STUB") == "1"  # launcher test on CPU: gloo + a stand-in decoder, never a measurement


def build_problem(N=144, p=0.003, rounds=12, W=3, F=1):
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.windows import plan_windows
    code, A, B = bb_code(N)
    dem = bb_dem(code, A, B, p, rounds)
    return plan_windows(dem.chk, dem.obs, dem.priors, N // 2, W, F, method=1)


def algorithmic_bytes(plan, stats, pre_max_iter, post_in_lds=False):
    """per BP iteration on a graph with E live edges, n live VNs, m live CNs and
    fp64 messages  B_iter = 40 E + 17 n + 2 m ; per window: sum over the iterations actually executed
    (full graph for the pre phase, shortened graph for the post phase) + sort 16 n per sort
    + OSD 2 * ceil((new_n+1)/64) * 8 bytes per GF(2) row addition applied + I/O (m + n).
    stats: int array [shots, W, 8] written by the kernel.  post_in_lds: leave the post-phase iterations out (the large-graph kernels keep
    the shortened graph's messages in LDS; counting them against HBM gave a fraction above 1)."""
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
        # (large-graph kernels: the shortened graph's messages live in LDS, only the full-graph iterations move their messages through HBM)
        b = pre_it * full + (0.0 if post_in_lds else post_it * short)
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


def reference_cpu_figure():
    """The reference's OWN compiled Cython osd_window timed beside the port on one core of the build container
    (tests/golden/time_reference.py -> profiles/r06_cpu_reference_vs_port.json; /root/reference does not exist on the GPU box, so
    the figure travels as a committed measurement): the GPU / CPU ratio of this line is against the port -- scale it by
    port_vs_reference to read it against the reference."""
    j = load_profile("r06_cpu_reference_vs_port.json")
    if not j:
        return {"reference_cython_per_core": None}
    return {"reference_cython_per_core": j["reference_cython_windows_per_s_per_core"],
            "port_per_core_same_run": j["port_windows_per_s_per_core"], "port_vs_reference": j["port_vs_reference"],
            "reference_figure_source": "profiles/r06_cpu_reference_vs_port.json (build container, " + str(j.get("cpu")) + ", 1 thread, "
                                       "recorded 192-shot sliding run, one decode() per window; SURVEY.md section 6 probe: 500-510 / core)"}


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
            "per_core": windows / busy / cores, **reference_cpu_figure()}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shots", type=int, default=None, help="shots per GPU per step (weak scaling); default 4096, bp4: 65536 decodes (a launch that fills the 256 CUs)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline",
                    help="headline = BASELINE configs[1] (default; the only one with roofline / cpu_baseline); gdg, bb288 = configs[2], [3]")
    ap.add_argument("--total-shots", type=int, default=4096 * 8, help="shots per step over all GPUs (strong scaling)")
    ap.add_argument("--osd-order", type=int, default=10, help="osd_cs order of the osd_window workloads (the notebooks' default is 10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-order0", "--no-order10", dest="no_side_order", action="store_true",
                    help="skip the extra measurement at the other OSD order (N = 1): order 0 next to the default 10, or 10 next to 0")
    ap.add_argument("--distinct-batches", type=int, default=4, help="pre-sampled batches cycled over the steps")
    ap.add_argument("--no-stream", action="store_true", help="time one launch at a time instead of the two-lane streaming entry")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip config.other_workloads (N = 1, headline): a few launches each of bb288, gdg, gdg64, global144 and bp4 after the timed region")
    args = ap.parse_args(argv)
    if args.shots is None:
        args.shots = default_shots(args.workload)
    return args


def default_shots(workload):
    """shots (decodes) per GPU per step: BASELINE's batch of 4096 for the sliding-window workloads; 65536 decodes for bp4 (a launch that
    fills the 256 CUs) and 2048 for the un-windowed model (one 1024-thread workgroup per decode, 2-8 ms each)"""
    return {"bp4": 65536, "global144": 2048}.get(workload, 4096)


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


class StubEngine:  # (launcher test only)
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

    def finish(self):
        pass


class GpuEngine:
    def __init__(self, args, rank, local_rank, lo, hi, plan, order, workload="headline", streaming=True):
        import torch
        from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
        self.torch = torch
        self.dev = torch.device("cuda", local_rank)
        self.plan, self.W = plan, len(plan.windows)
        kw = dict(GDG_KW) if workload == "gdg" else (dict(GDG64_KW) if workload == "gdg64" else
                                                     dict(DECODER_KW, osd_order=order, **WORKLOADS[workload].get("decoder_kw", {})))
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
        # one set of output buffers per lane of the stream object (a launch writes the set of its lane)
        self.streaming = streaming
        self.out = [dict(total=torch.empty((shots, plan.chk.shape[1]), dtype=torch.uint8, device=self.dev),
                         stats=torch.empty((shots, self.W, 8), dtype=torch.int32, device=self.dev),
                         shot_result=torch.empty((shots, 2), dtype=torch.int32, device=self.dev)) for _ in range(2 if streaming else 1)]
        self.stream = self.dec.stream(shots) if streaming else None
        self.nstep = 0
        self.last = 0
        if streaming:  # one launch per lane before anything is timed (the first launch on a lane creates its hardware queue and its buffers)
            for k in range(2):
                self.stream.push_device(self.dets[0], min_pm=None, **self.out[k])
            self.stream.wait()

    @property
    def shot(self):  # decisions of the most recent step
        return self.out[self.last]["shot_result"]

    @property
    def stats(self):
        return self.out[self.last]["stats"]

    def step(self, i):
        if self.streaming:
            self.last = self.nstep & 1
            self.stream.push_device(self.dets[i % self.nb], min_pm=None, **self.out[self.last])
            self.nstep += 1
        else:
            self.dec.decode_device(self.dets[i % self.nb], min_pm=None, **self.out[0])

    def finish(self):
        """the current torch stream waits for both lanes (what follows on it -- the gather of the decisions -- sees the results)"""
        if self.streaming:
            self.stream.wait(self.torch.cuda.current_stream(self.dev))

    def kernel_timing(self, i0, k):
        """k launches one at a time with HIP events around each on the launch stream (swd_pipeline_set_timing): the single-launch
        kernel duration the roofline and the committed rocprof profile describe -> (total ms, launches, wall seconds)"""
        o = self.out[0]
        self.sync()
        for _ in range(4):  # one untimed launch per launch slot of the handle: a form the streamed steps did not use allocates its scratch now
            self.dec.decode_device(self.dets[i0 % self.nb], min_pm=None, **o)
        self.sync()
        self.dec.set_timing(True)
        t0 = time.perf_counter()
        for i in range(k):
            self.dec.decode_device(self.dets[(i0 + i) % self.nb], min_pm=None, **o)
        self.sync()
        wall = time.perf_counter() - t0
        ms, n = self.dec.get_timing()
        self.dec.set_timing(False)
        return ms, n, wall

    def check_status(self):
        self.dec.check_status()  # raises if any window of any launch gave up waiting for its predecessor

    def sync(self):
        if self.streaming:
            self.stream.wait()
        self.torch.cuda.synchronize()


class Bp4Engine:
    """bp4_osd on device-resident syndromes (one launch per step); decisions = exit word + iteration count per shot.
    streaming: consecutive steps go round four HIP streams (swd_bp4_decode_batch_dev takes the stream; the handle's four launch
    slots keep the launches' scratch apart) and four sets of output buffers -- a launch ends on the few decodes that run all
    max_iter iterations (0.9 ms of a 1.8 ms launch with one workgroup busy), the other launches' grids fill that tail."""

    def __init__(self, args, rank, local_rank, lo, hi, streaming=False):
        import torch
        from slidingwindowdecoder_amd import bp4_osd
        from slidingwindowdecoder_amd.codes import bb_code
        self.torch, self.W, self.kernel_ms, self.launches = torch, 1, 0.0, 0
        self.dev = torch.device("cuda", local_rank)
        code, _, _ = bb_code(144)
        n, p = code.hx.shape[1], 0.02
        self.edges = int(np.count_nonzero(code.hx)) + int(np.count_nonzero(code.hz))
        pr = np.full(n, p / 3)
        self.dec = bp4_osd(code.hx, code.hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100,
                           ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10, device=local_rank)
        shots = hi - lo
        self.nb = max(1, min(args.distinct_batches, args.steps + args.warmup))
        self.sx, self.sz = [], []
        for i in range(self.nb):  # Pauli errors keyed by (batch, global first shot): a shot's data does not depend on the number of ranks
            rng = np.random.default_rng([20240318, i, lo])
            pauli = rng.choice(4, size=(shots, n), p=[1 - p, p / 3, p / 3, p / 3])  # 0 I, 1 X, 2 Y, 3 Z
            ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
            self.sx.append(torch.from_numpy(np.ascontiguousarray((ez @ code.hx.T % 2).astype(np.uint8))).to(self.dev))
            self.sz.append(torch.from_numpy(np.ascontiguousarray((ex @ code.hz.T % 2).astype(np.uint8))).to(self.dev))
        self.streaming = streaming
        # four launches in flight: the handle has four launch slots, and a launch of 65 536 decodes ends on decodes that take as long as
        # its whole body ([[144]]: 1 / 2 / 3 / 4 launches in flight 35 / 66 / 70 / 73 M decodes/s, scripts/bp4_lanes.py)
        self.nl = 4 if streaming else 1
        self.outs = [torch.empty((shots, 2, n), dtype=torch.uint8, device=self.dev) for _ in range(self.nl)]
        self.stat = [torch.empty((shots, 8), dtype=torch.int32, device=self.dev) for _ in range(self.nl)]
        # (streams of both priorities: streams of one priority may share a hardware queue, and those that do run back to back -- swd_osdw.hip)
        self.lanes = [torch.cuda.Stream(self.dev, priority=-(i & 1)) for i in range(self.nl)] if streaming else None
        for k, lane in enumerate(self.lanes or []):  # the first launch on a stream creates its hardware queue: not inside a timed step
            lane.wait_stream(torch.cuda.current_stream(self.dev))
            self.dec.decode_batch_device(self.sx[0], self.sz[0], out=self.outs[k], stats=self.stat[k], stream=lane)
        torch.cuda.synchronize()
        self.nstep = self.last = 0
        self.timing, self.events = False, []

    @property
    def stats(self):  # of the most recent step
        return self.stat[self.last]

    @property
    def shot(self):
        return self.stat[self.last][:, :2]

    def step(self, i):
        if self.streaming and not self.timing:
            self.last = self.nstep % self.nl
            self.nstep += 1
            lane = self.lanes[self.last]
            if self.nstep <= self.nl:  # the syndromes were produced on the current stream
                lane.wait_stream(self.torch.cuda.current_stream(self.dev))
            self.dec.decode_batch_device(self.sx[i % self.nb], self.sz[i % self.nb], out=self.outs[self.last], stats=self.stat[self.last], stream=lane)
            return
        self.last = 0
        if self.timing:  # HIP events on the launch stream (torch's current stream is the stream the kernel is launched on)
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
        self.dec.decode_batch_device(self.sx[i % self.nb], self.sz[i % self.nb], out=self.outs[0], stats=self.stat[0])
        if self.timing:
            e1.record()
            self.events.append((e0, e1))

    def set_timing(self, on):
        self.timing = on

    def get_timing(self):
        self.torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        k = len(self.events)
        self.events = []
        return ms, k

    def check_status(self):
        pass

    def finish(self):
        """the current torch stream waits for both lanes (the gather of the decisions follows on it)"""
        if self.streaming:
            for lane in self.lanes:
                self.torch.cuda.current_stream(self.dev).wait_stream(lane)

    def kernel_timing(self, i0, k):
        self.sync()
        self.set_timing(True)
        for i in range(4):  # untimed: the single-launch form (start-order kernels, one stream) settles after the streamed steps
            self.step(i0 + i)
        self.sync()
        self.events = []
        t0 = time.perf_counter()
        for i in range(k):
            self.step(i0 + i)
        self.sync()
        wall = time.perf_counter() - t0
        ms, n = self.get_timing()
        self.set_timing(False)
        return ms, n, wall

    def sync(self):
        self.torch.cuda.synchronize()


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def find_profile(workload):
    """The committed counter profile of this workload's kernel: (sq counters, summary, source names), newest tag first.
    The round-2 files carry no workload in their names and belong to the headline kernel at OSD order 0."""
    for tag in PROFILE_TAGS:
        stem = f"{tag}_{workload}"
        sq, sm = load_profile(f"{stem}_sq_counters.json"), load_profile(f"{stem}_summary.json")
        if sq and sm:
            return sq, sm, f"profiles/{stem}_sq_counters.json", f"profiles/{stem}_summary.json"
    if workload == "headline":
        sq, sm = load_profile("r02_sq_counters.json"), load_profile("r02_summary.json")
        if sq and sm:
            return sq, sm, "profiles/r02_sq_counters.json", "profiles/r02_summary.json"
    return None, None, None, None


def find_stream_profile(workload):
    """The committed kernel trace of this command in its STREAMED step mode (scripts/profile_stream.sh ->
    profiles/<tag>_<workload>_stream_summary.json): per-launch durations under overlap, the overlap fraction, the makespan per step."""
    for tag in PROFILE_TAGS:
        name = f"{tag}_{workload}_stream_summary.json"
        j = load_profile(name)
        if j:
            keep = ("launches", "avg_kernel_ms_under_overlap", "makespan_ms_per_step", "overlap_fraction", "avg_concurrent_launches", "git")
            return dict({k: j.get(k) for k in keep}, source="profiles/" + name)
    return None


def lds_algorithmic_bytes(plan, stats):
    """LDS bytes the BP iterations have to move: every live edge's message is read and written once by each of the
    two passes = 32 bytes per live edge and executed iteration (full graph in the pre phase, the shot's own live edges
    in the post phase; both counts come from the kernel's statistics)."""
    total = 0.0
    for wi, w in enumerate(plan.windows):
        st = stats[:, wi, :].astype(np.float64)
        total += (32.0 * w.mat.nnz * st[:, 2] + 32.0 * st[:, 6] * st[:, 3]).sum()
    return total


def gdg_lds_algorithmic_bytes(plan, stats, new_n=None):
    """Guessing decoders (statistics words of include/swd.h: [2] pre-processing iterations on the full graph, [3] iterations inside
    decimation steps on the sub-graph of the new_n = min(n, 2m) least reliable columns, whose live edges shrink with every decimation and
    are not reported): -> (lower, upper) bound of 32 B per live edge and iteration.  lower = the pre-processing iterations alone; upper
    adds every decimation-step iteration at the edge count of the heaviest new_n columns of the window."""
    lo = up = 0.0
    for wi, w in enumerate(plan.windows):
        st = stats[:, wi, :].astype(np.float64)
        m, n = w.mat.shape
        k = min(n, 2 * m)
        deg = np.sort(np.asarray(w.mat.sum(axis=0)).ravel())[::-1]
        e_sub = float(deg[:k].sum())
        lo += (32.0 * w.mat.nnz * st[:, 2]).sum()
        up += (32.0 * w.mat.nnz * st[:, 2] + 32.0 * e_sub * st[:, 3]).sum()
    return lo, up


def bp4_lds_algorithmic_bytes(dec_edges, stats):
    """bp4_osd: both Tanner graphs' messages are read and written once by each of the two passes of every executed iteration
    (statistics word [1] = bp_iteration): 32 B x (nnz(Hx) + nnz(Hz)) per iteration."""
    return 32.0 * dec_edges * float(stats[:, 1].astype(np.float64).sum())


LDS_MIX_PEAK_GBS = 79000.0  # what a half-read / half-write ds_*_b64 mix can move with every CU streaming (MI355X_MICROARCH.md, LDS: ~150 TB/s reads, 38-51 TB/s writes)


def roofline(workload, kernel, alg_bytes, lds_alg_bytes, avg_kernel_s, irreducible, step_s=None, step_mode=None, lds_alg_note=None):
    """`frac` = achieved / peak of the memory that binds the kernel, from ALGORITHMIC bytes and the kernel time measured live with
    HIP events: LDS-resident kernels -> lds_alg_bytes / t / 79 TB/s (bound "lds"); a kernel whose messages live in HBM
    (lds_alg_bytes None, alg_bytes given: global144) -> SURVEY 8(d)'s bytes / t / 8 TB/s (bound "hbm").
    `utilisation` = the busy fractions of the units the kernel could saturate, from the committed per-launch means of separate
    rocprofv3 --pmc passes of this same command (profiles/), each <= 1 by construction -- they rise with bank conflicts and
    padding, so they describe how occupied a unit is, never how good the kernel is:
      lds   (SQ_LDS_IDX_ACTIVE + 2 x SQ_INSTS_LDS_STORE) / (256 CUs x 2.4 GHz x t): cycles the CU's LDS pipeline is occupied --
            array cycles incl. bank conflicts, plus the two cycles by which the address / data transfer of a store exceeds its
            array cycles (guide, LDS table: ds_write_b32 4 vs 2, ds_write_b64 6 vs 4)
      valu  (4 x fp64-rate instructions + 2 x the other VALU instructions) / (1024 SIMDs x 2.4 GHz x t); fp64-rate = the hardware's
            SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 + the min / max / compare instructions on doubles, estimated as ADD_F64 x their
            static ratio in the kernel's ISA (profiles/<tag>_<workload>_isa_mix.json) -- no counter separates them
      hbm   (2 x FETCH_SIZE + WRITE_SIZE) / (8 TB/s x t)"""
    sq, sm, sq_src, sm_src = find_profile(workload)
    out = {"kernel": kernel, "avg_kernel_ms": avg_kernel_s * 1e3, "bound": "lds", "frac": None, "achieved": None,
           "peak": None, "unit": None, "traffic": None}
    fr, diag = {}, {}
    if sq and sm:
        c = sq["per_launch_mean"]
        # (bp4: a batch is two launches -- BP, then OSD on the queue of unconverged decodes -- and the HIP events bracket both)
        prof_ms = sm.get("avg_ms_with_companion", sm.get("avg_ms"))
        out["profile"] = {"counters": sq_src, "kernel_stats": sm_src, "profiled_kernel": sm.get("kernel"), "profiled_avg_kernel_ms": prof_ms,
                          "git": sm.get("git")}
        if "companion_kernel" in sm:
            out["profile"]["companion_kernel"] = sm["companion_kernel"]
            out["profile"]["companion_avg_ms"] = sm.get("companion_avg_ms")
        # the counters belong to the binary that was profiled: flag the line when the live kernel time has moved away from it
        out["profile_stale"] = bool(prof_ms and abs(avg_kernel_s * 1e3 - prof_ms) > 0.05 * prof_ms)
        if "SQ_LDS_IDX_ACTIVE" in c:
            stores = c.get("SQ_INSTS_LDS_STORE")
            cyc = c["SQ_LDS_IDX_ACTIVE"] + (2.0 * stores if stores is not None else 0.0)
            busy = cyc / NUM_CU / PEAK_CLOCK_HZ
            fr["lds"] = {"frac": busy / avg_kernel_s, "busy_ms_at_peak_clock": busy * 1e3,
                         "array_only_frac": c["SQ_LDS_IDX_ACTIVE"] / NUM_CU / PEAK_CLOCK_HZ / avg_kernel_s,
                         "bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"],
                         "counter": "(SQ_LDS_IDX_ACTIVE + 2 x SQ_INSTS_LDS_STORE) / (256 CUs x 2.4 GHz)" if stores is not None
                                    else "SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz) (no per-kind LDS pass in this profile)"}
            if stores is not None:
                fr["lds"]["instructions_load_store_atomic"] = [c.get("SQ_INSTS_LDS_LOAD"), stores, c.get("SQ_INSTS_LDS_ATOMIC")]
                for k in ("SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"):
                    if k in c:
                        fr["lds"][k.lower() + "_share_of_cu_cycles"] = c[k] / NUM_CU / PEAK_CLOCK_HZ / avg_kernel_s
        if "SQ_INSTS_VALU" in c:
            mix = load_profile(sq_src.split("/")[-1].replace("_sq_counters.json", "_isa_mix.json")) if sq_src else None
            if "SQ_INSTS_VALU_ADD_F64" in c:
                hw64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
                ratio = (mix or {}).get("f64_minmaxcmp_per_f64_add")
                est = c["SQ_INSTS_VALU_ADD_F64"] * ratio if ratio is not None else None
                f64 = min(hw64 + (est or 0.0), c["SQ_INSTS_VALU"])
                cyc = 4.0 * f64 + 2.0 * (c["SQ_INSTS_VALU"] - f64)
                busy = cyc / (NUM_CU * SIMD_PER_CU) / PEAK_CLOCK_HZ
                fr["valu"] = {"frac": busy / avg_kernel_s, "busy_ms_at_peak_clock": busy * 1e3,
                              "fp64_rate_instructions": f64, "fp64_rate_counted_by_hardware": hw64, "fp64_minmaxcmp_estimated": est,
                              "all_valu_instructions": c["SQ_INSTS_VALU"],
                              "bounds": {"every_instruction_at_2_cycles": 2.0 * c["SQ_INSTS_VALU"] / (NUM_CU * SIMD_PER_CU) / PEAK_CLOCK_HZ / avg_kernel_s,
                                         "every_instruction_at_4_cycles": 4.0 * c["SQ_INSTS_VALU"] / (NUM_CU * SIMD_PER_CU) / PEAK_CLOCK_HZ / avg_kernel_s},
                              "counter": "(4 x fp64-rate + 2 x other VALU instructions) / (1024 SIMDs x 2.4 GHz)"
                                         + ("" if ratio is not None else " -- fp64 min / max / compare NOT included (no ISA mix file): lower bound")}
            else:  # an older profile without the per-type pass: the bracket only
                lo = 2.0 * c["SQ_INSTS_VALU"] / (NUM_CU * SIMD_PER_CU) / PEAK_CLOCK_HZ
                fr["valu"] = {"frac": 2.0 * lo / avg_kernel_s, "busy_ms_at_peak_clock": 2.0 * lo * 1e3,
                              "bounds": {"every_instruction_at_2_cycles": lo / avg_kernel_s, "every_instruction_at_4_cycles": 2.0 * lo / avg_kernel_s},
                              "counter": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz): UPPER bound, no per-type pass in this profile"}
        traffic = sm.get("hbm_bytes_per_launch")
        if traffic:
            out["traffic"] = traffic
            fr["hbm"] = {"frac": traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS, "GBps": traffic / avg_kernel_s / 1e9,
                         "counter": "(2 x FETCH_SIZE + WRITE_SIZE) per launch / 8 TB/s"}
        # diagnostics that are NOT capacities (sums of wave time: several waves of a CU can be inside an LDS instruction at once)
        if "SQ_ACTIVE_INST_LDS" in c:
            diag["lds_pipe"] = {"value": c["SQ_ACTIVE_INST_LDS"] * 4.0 / NUM_CU / PEAK_CLOCK_HZ / avg_kernel_s,
                                "meaning": "wave time inside LDS instructions per CU and launch time (SQ_ACTIVE_INST_LDS x 4 / (256 CUs x 2.4 GHz x t)); unbounded"}
        if c.get("SQ_WAVE_CYCLES"):
            diag["wave_cycles"] = {"issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                                   "parked_waitcnt_or_barrier": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                                   "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]}
        if "SQ_INSTS_LDS" in c:
            out["lds_bytes_moved"] = c["SQ_INSTS_LDS"] * 512.0
            out["lds_bytes_moved_note"] = "SQ_INSTS_LDS x 512 B (one 8-byte access per lane of a wave instruction)"
        if sm.get("dispatch"):
            out["scratch_bytes_per_lane"] = int(sm["dispatch"].get("Scratch_Size", 0) or 0)
    if lds_alg_bytes is not None:
        out["lds_bytes_algorithmic"] = lds_alg_bytes
        out["lds_algorithmic_frac"] = lds_alg_bytes / avg_kernel_s / 1e9 / LDS_MIX_PEAK_GBS
        if out.get("lds_bytes_moved"):
            out["lds_padding_factor"] = out["lds_bytes_moved"] / lds_alg_bytes
    if irreducible is not None:
        out["irreducible_hbm_bytes"] = irreducible
        out["traffic_over_irreducible"] = (out["traffic"] / irreducible) if out["traffic"] else None
    if alg_bytes is not None:
        out.update({"algorithmic_bytes_per_launch": alg_bytes, "achieved_algorithmic": alg_bytes / avg_kernel_s / 1e9,
                    "achieved_algorithmic_unit": "GB/s",
                    "achieved_algorithmic_over_hbm_peak": alg_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBS})
    util = dict(fr)
    if fr:
        busiest = max(fr, key=lambda k: fr[k]["frac"])
        util["busiest"] = busiest
        util["max"] = fr[busiest]["frac"]
        util["note"] = ("busy share of a unit's capacity at the 2.4 GHz peak clock (LDS pipeline cycles incl. bank conflicts and store "
                        "transfers; VALU instructions priced by width; HBM bytes): occupancy, not achieved / peak")
    out["utilisation"] = util
    out["diagnostics"] = diag
    if fr and step_s:
        # The same counters priced at the STEP time of the timed region (this rank's shots per step): with the two-lane stream,
        # consecutive launches overlap -- a launch's grid fills the workgroup slots the previous launch's tail leaves empty -- so a step
        # takes less than one launch does alone (ms_per_step < avg_kernel_ms) and the device does one launch's work per step.
        sc = avg_kernel_s / step_s
        out["at_step_time"] = {"ms_per_step": step_s * 1e3, "avg_kernel_ms_single_launch": avg_kernel_s * 1e3, "step_mode": step_mode,
                               "utilisation": {k: v["frac"] * sc for k, v in fr.items()},
                               "note": "one launch's counters and algorithmic bytes over the step time of the timed region; `frac` above "
                                       "is priced at the single-launch kernel time that the committed kernel-trace describes"}
        ss = find_stream_profile(workload)
        if ss:
            out["at_step_time"]["streamed_profile"] = ss
    # achieved / peak of the binding memory, from algorithmic bytes
    if lds_alg_bytes is not None:
        ach = lds_alg_bytes / avg_kernel_s / 1e9
        out.update({"bound": "lds", "achieved": ach, "peak": LDS_MIX_PEAK_GBS, "unit": "GB/s", "frac": ach / LDS_MIX_PEAK_GBS,
                    "achieved_is": "32 B per live edge and executed BP iteration (kernel statistics) / kernel time"
                                   + (" -- " + lds_alg_note if lds_alg_note else ""),
                    "peak_is": "LDS, half-read / half-write 8-byte mix with all 256 CUs streaming (MI355X_MICROARCH.md, LDS section)"})
    elif alg_bytes is not None:
        ach = alg_bytes / avg_kernel_s / 1e9
        out.update({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "achieved_is": "SURVEY 8(d): 40E+17n+2m per executed FULL-GRAPH BP iteration (the shortened graph's messages are in LDS) + sort + OSD row additions + I/O (kernel statistics) / kernel time",
                    "peak_is": "HBM3E 8 TB/s (spec)"})
    if out.get("frac") is not None and step_s:
        out["at_step_time"] = dict(out.get("at_step_time") or {"ms_per_step": step_s * 1e3, "step_mode": step_mode},
                                   frac=out["frac"] * avg_kernel_s / step_s)
    out["note"] = ("frac = achieved / peak with ALGORITHMIC bytes: the messages of this kernel live in "
                   + ("LDS, so the binding memory is LDS (SURVEY 8(d)'s HBM-priced figure is kept as achieved_algorithmic: it exceeds the "
                      "HBM peak, which is the traffic the LDS-resident design avoids); " if lds_alg_bytes is not None else "HBM; ")
                   + "utilisation = busy fractions from the committed counters (not a score); time measured here with HIP events on single launches")
    return out


def measure_other_workload(workload, args, rank, local_rank, steps=5, shots=None):
    """One of the other workloads on this GPU after the headline's timed region: `steps` steps in the step mode `bench.py --workload
    <w>` uses (after one warm-up step), then three single launches under HIP events -> a compact record for config.other_workloads."""
    import copy
    wl = WORKLOADS[workload]
    a = copy.copy(args)
    a.workload, a.shots, a.steps, a.warmup = workload, (shots or default_shots(workload)), steps, 1
    streaming = not args.no_stream
    t_setup = time.perf_counter()
    if workload == "bp4":
        eng = Bp4Engine(a, rank, local_rank, 0, a.shots, streaming=streaming)
    else:
        plan = build_problem(**wl["problem"])
        eng = GpuEngine(a, rank, local_rank, 0, a.shots, plan, args.osd_order, workload, streaming=streaming)
    eng.step(0)
    eng.sync()
    t_setup = time.perf_counter() - t_setup
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(1 + i)
    eng.finish()
    eng.sync()
    el = time.perf_counter() - t0
    eng.check_status()
    st = eng.stats.cpu().numpy()
    ms, n, _ = eng.kernel_timing(1 + steps, 3)
    eng.check_status()
    avg_s = ms / max(n, 1) / 1e3
    alg = lds_alg = None
    note = None
    extra = {}
    if workload == "bp4":
        lds_alg = bp4_lds_algorithmic_bytes(eng.edges, st)
    elif workload == "global144":
        alg = algorithmic_bytes(plan, st, None, post_in_lds=True)
    elif workload in ("gdg", "gdg64"):
        lds_alg, up = gdg_lds_algorithmic_bytes(plan, st)
        note = "LOWER bound: the pre-processing iterations only (the live edges of the decimation steps are not in the statistics)"
        extra["roofline_frac_upper_bound"] = up / avg_s / 1e9 / LDS_MIX_PEAK_GBS
    else:
        lds_alg = lds_algorithmic_bytes(plan, st)
    r = roofline(workload, "swd::bp4_kernel" if workload == "bp4" else "swd::pipeline_kernel", alg, lds_alg, avg_s, None, lds_alg_note=note)
    cls = np.bincount((st[..., 0] & 0xFF).ravel(), minlength=7)
    rec = {"workload": workload, "metric": wl["metric"], "value": a.shots * eng.W * steps / el, "unit": wl.get("unit", "windows/s"),
           "shots_per_step": a.shots, "steps": steps, "ms_per_step": el / steps * 1e3,
           "step_mode": ("four HIP streams in turn" if workload == "bp4" else "two-lane stream") if streaming else "one launch at a time",
           "ms_per_launch": avg_s * 1e3, "launches_timed": int(n), "exit_classes": [int(x) for x in cls[:7]],
           "roofline_bound": r["bound"], "roofline_frac": r["frac"], "roofline_achieved_GBps": r["achieved"], "roofline_peak_GBps": r["peak"],
           "lds_algorithmic_frac": r.get("lds_algorithmic_frac"),
           "utilisation_busiest": r["utilisation"].get("busiest"), "utilisation_max": r["utilisation"].get("max"),
           "scratch_bytes_per_lane": r.get("scratch_bytes_per_lane"),
           "profile": (r.get("profile") or {}).get("counters"),
           "profile_stale": r.get("profile_stale"), "setup_s": t_setup}
    if note:
        rec["roofline_frac_note"] = note
    if shots is not None and shots != default_shots(workload):  # (the committed profile describes launches of the default batch size)
        rec["profile_stale"] = None
        rec["profile_note"] = "counters profiled at the workload's default batch size; the launch time of this batch size is not comparable"
    rec.update(extra)
    return rec


def time_steps(engine, args, dist, world, total_shots):
    """W untimed steps, then exactly K timed steps + the gather of the decisions, bracketed by barriers and
    device synchronisation; returns (elapsed seconds = max over ranks, gathered decisions, per-rank record).
    The per-rank record splits every rank's timed region into its K steps (launch .. device idle) and the gather that follows, so
    that a multi-GPU line shows imbalance by itself: ms_per_step min / max over ranks and the gather time."""
    from slidingwindowdecoder_amd.distributed import gather_decisions
    import torch
    for i in range(args.warmup):
        engine.step(i)
    engine.sync()
    if dist.is_initialized():
        dist.barrier()
    engine.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        engine.step(args.warmup + i)
    engine.finish()  # (streaming: the current stream waits for both lanes before the decisions are gathered)
    engine.sync()
    t_steps = time.perf_counter() - t0
    gathered = gather_decisions(engine.shot, total_shots)  # per-shot decisions of the last step, over RCCL
    engine.sync()
    t_gather = time.perf_counter() - t0 - t_steps
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    mine = [t_steps, t_gather, elapsed]
    per_rank = [mine]
    if dist.is_initialized():
        t = torch.tensor(mine, dtype=torch.float64, device=engine.shot.device)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [[float(x) for x in a.cpu()] for a in allt]
        elapsed = max(r[2] for r in per_rank)
    k = max(args.steps, 1)
    rec = {"ms_per_step_per_rank": [r[0] / k * 1e3 for r in per_rank],
           "ms_per_step_min_over_ranks": min(r[0] for r in per_rank) / k * 1e3,
           "ms_per_step_max_over_ranks": max(r[0] for r in per_rank) / k * 1e3,
           "gather_ms_per_rank": [r[1] * 1e3 for r in per_rank],
           "gather_ms_max_over_ranks": max(r[1] for r in per_rank) * 1e3,
           "note": "each rank's K timed steps (first launch .. device idle) and its all_gather of the decisions, host clock; "
                   "`ms_per_step` of the line = (max over ranks of steps + gather + closing barrier) / K"}
    return elapsed, gathered, rec


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the job has WORLD_SIZE={world}")
    # a process group exists whenever the job was started by torch.distributed.run -- also with one rank, so that the RCCL
    # initialisation, the barrier and the all_gather of the decisions are the code that runs at every N (SWD_BENCH_DIST=0/1 overrides)
    use_dist = os.environ.get("SWD_BENCH_DIST", "1" if ("WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ) else "0") == "1" or world > 1

    wl = WORKLOADS[args.workload]
    headline = args.workload == "headline"
    osdw = args.workload in ("headline", "bb288", "global144")
    # the two-lane stream is the step mode where overlapping consecutive launches pays: the osd_window workloads and gdg() -- the next
    # launch's grid fills the tail of the previous one, and a stream's gdg() batches take the serial tree walk (1.19 -> 1.56 M windows/s,
    # round 6) and so do the threaded ensemble's (tickets instead of the work-item ring: +3-6 % at 4096 shots, +9 % at 16 384).
    # bp4_osd: four HIP streams handed to swd_bp4_decode_batch_dev in turn (a launch ends on the few decodes that run all max_iter iterations)
    streaming = not args.no_stream and not STUB
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not STUB and headline:
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
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (only a one-process job can get here without one: any free port will do)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus

    total_shots = args.shots * world if args.scaling == "weak" else args.total_shots
    lo, hi = shard_bounds(total_shots, rank, world)
    per_rank = [b - a for a, b in (shard_bounds(total_shots, r, world) for r in range(world))]
    plan = None if (STUB or wl["problem"] is None) else build_problem(**wl["problem"])
    if STUB:
        engine = StubEngine(args, rank, lo, hi)
    elif args.workload == "bp4":
        engine = Bp4Engine(args, rank, local_rank, lo, hi, streaming=streaming)
    else:
        engine = GpuEngine(args, rank, local_rank, lo, hi, plan, args.osd_order, args.workload, streaming=streaming)
    W = engine.W
    elapsed, gathered, rank_times = time_steps(engine, args, dist, world, total_shots)
    assert gathered.shape[0] == total_shots, (gathered.shape, total_shots)
    backend = dist.get_backend() if dist.is_initialized() else None
    dist_ranks = dist.get_world_size() if dist.is_initialized() else 0

    line = {
        "metric": wl["metric"],
        "value": total_shots * W * args.steps / elapsed,
        "unit": wl.get("unit", "windows/s"),
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
                          "shots_this_rank": hi - lo, "shots_per_rank": per_rank, "gather_ok": bool(ok),
                          "collective_backend": backend, "collective_ranks": dist_ranks, "rank_times": rank_times}
        if rank == 0:
            print(json.dumps(line))
        if dist.is_initialized():
            dist.destroy_process_group()
        if not ok:
            raise SystemExit("gathered decisions differ from the expected ones")
        return

    engine.check_status()
    # accounting data of the LAST timed step (copied before the timing loop below re-uses the first lane's buffers)
    st = engine.stats.cpu().numpy()
    sr = engine.shot.cpu().numpy()
    # the single-launch kernel duration behind the roofline: HIP events around each launch on the launch stream, one launch at a
    # time, after the timed region (streamed launches overlap, so their individual durations would not describe the kernel)
    kt = max(3, min(args.steps, 20))
    kern_ms, launches, single_wall = engine.kernel_timing(args.warmup + args.steps, kt)
    engine.check_status()

    # accounting (outside the timed region), on this rank's shard of the last step
    avg_kernel_s = kern_ms / max(launches, 1) / 1e3
    cfg = {
        "workload": (wl["desc"] % args.osd_order if "%d" in wl["desc"] else wl["desc"]) + ", "
                    + ("".join(f"DIAGNOSTIC RUN {k}={DECODER_KW[k]}, " for k in ("pre_max_iter", "post_max_iter") if os.environ.get("SWD_BENCH_" + k.upper())))
                    + (f"{args.shots} shots per GPU per step" if args.scaling == "weak" else f"{total_shots} shots per step split over the GPUs"),
        "world_size": world, "shots_total": total_shots, "shots_rank0": hi - lo, "shots_per_rank": per_rank, "windows_per_shot": W,
        "parallelism": f"shots sharded over {world} GPU(s), no data-path collective; one all_gather of 8 B per shot",
        # what closed the timed region: the RCCL (backend nccl) all_gather over this many ranks, or nothing (plain one-process run)
        "collective_backend": backend, "collective_ranks": dist_ranks, "rank_times": rank_times,
        "kernel_launches_timed": int(launches),
        "step_mode": (("four HIP streams in turn (swd_bp4_decode_batch_dev): consecutive steps overlap" if args.workload == "bp4" else
                       "two-lane stream (swd_pipeline_stream_push_dev): consecutive steps overlap") if streaming else "one launch at a time"),
        "single_stream_windows_per_s": (hi - lo) * W * launches / single_wall if launches else None,
        "single_stream_note": "this rank's shots, one launch at a time with HIP events and a host synchronisation per launch (the loop that times the kernel)",
    }
    alg_bytes = lds_alg = irr = lds_note = None
    if args.workload == "bp4":
        cls = np.bincount((st[:, 0] & 0xFF).ravel(), minlength=7)
        cfg["exit_classes_bp_osd_rank0"] = [int(cls[0]), int(cls[2])]
        cfg["converged_fraction_rank0"] = float(((st[:, 0] & 0x100) != 0).mean())
        kernel = "swd::bp4_kernel"
        lds_alg = bp4_lds_algorithmic_bytes(engine.edges, st)
    else:
        last = (args.warmup + args.steps - 1) % engine.nb
        logical = (sr[:, 0].astype(np.int64) != engine.obs_true[last]) | (sr[:, 1] != 0)
        cls = np.bincount((st[..., 0] & 0xFF).ravel(), minlength=7)
        cfg.update({"exit_classes_pre_post_osd_rank0": [int(cls[0]), int(cls[1]), int(cls[2])], "sched_faults": int(cls[6]),
                    "logical_errors_last_step_rank0": int(logical.sum())})
        kernel = "swd::pipeline_kernel"
        irr = irreducible_hbm_bytes(plan, hi - lo)
        if osdw:
            cfg["bp_iterations_pre_post_rank0"] = [int(st[..., 2].sum()), int(st[..., 3].sum())]
            cfg["live_edge_iterations_pre_post_rank0"] = [int(sum(w.mat.nnz * st[:, i, 2].sum() for i, w in enumerate(plan.windows))),
                                                          int((st[..., 6].astype(np.int64) * st[..., 3]).sum())]
            alg_bytes = algorithmic_bytes(plan, st, DECODER_KW["pre_max_iter"], post_in_lds=args.workload == "global144")
            lds_alg = lds_algorithmic_bytes(plan, st) if args.workload != "global144" else None  # (its messages are not in LDS)
        else:
            lds_alg, lds_up = gdg_lds_algorithmic_bytes(plan, st)
            lds_note = "LOWER bound: the pre-processing iterations only (the live edges of the decimation steps are not in the statistics)"
            cfg["lds_bytes_algorithmic_upper_bound"] = lds_up

    if rank == 0 and world == 1 and headline and not args.no_side_order:
        # the same batches at the other OSD order (a second, untimed-by-the-driver loop): order 0 next to the notebooks' default 10
        side = 0 if args.osd_order != 0 else 10
        e2 = GpuEngine(args, rank, local_rank, lo, hi, plan, side, streaming=streaming)
        k2 = max(2, min(args.steps, 6))
        e2.step(0); e2.sync()
        t0 = time.perf_counter()
        for i in range(k2):
            e2.step(i)
        e2.sync()
        cfg[f"osd_cs_order{side}_windows_per_s"] = total_shots * W * k2 / (time.perf_counter() - t0)  # (same step mode)
        e2.check_status()

    if rank == 0 and world == 1 and headline and not args.no_other_workloads:
        # the other five workloads, a few launches each, so that the driver's record carries every rate this repository claims
        # (each has its own `bench.py --workload <w>` line and committed counter profile; here: value, ms per launch, staleness)
        del engine
        torch.cuda.empty_cache()
        cfg["other_workloads"] = []
        # (the guessing decoders a second time at 16384 shots per launch: at 4096 their launches are bound by the longest shots' chains of
        #  eleven windows, not by the device -- BASELINE configs[2] names no batch size)
        for w, sh in (("bb288", None), ("gdg", None), ("gdg", 16384), ("gdg64", None), ("gdg64", 16384), ("global144", None), ("bp4", None)):
            try:  # a failing side workload must not lose the headline record
                # (streamed workloads: the last step's tail is not overlapped by a next step -- enough steps to amortise it)
                k = {"gdg": 20, "gdg64": 12, "bp4": 30, "global144": 10}.get(w, 5) if sh is None else (6 if w == "gdg" else 3 if w == "gdg64" else 2)
                cfg["other_workloads"].append(measure_other_workload(w, args, rank, local_rank, steps=k, shots=sh))
            except Exception as e:  # noqa: BLE001
                cfg["other_workloads"].append({"workload": w, "error": f"{type(e).__name__}: {e}"[:500]})
            torch.cuda.empty_cache()

    if rank == 0:
        line["config"] = cfg
        # (per-GPU step time: under weak scaling every rank runs the same number of shots per step)
        line["roofline"] = roofline(args.workload, kernel, alg_bytes, lds_alg, avg_kernel_s, irr, step_s=elapsed / args.steps,
                                    step_mode=cfg["step_mode"], lds_alg_note=lds_note)
        line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
