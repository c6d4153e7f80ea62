#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one MI355X (bench.py measures configs[1]):

  config 3  [[144,12,12]] circuit-level p=0.003, (W,F)=(3,1), bpgdg_decoder windows (guessing.py:160-173)
  config 4  [[288,12,18]] circuit-level p=0.003, (W,F)=(4,1), osd_window windows
  4gdg      [[288,12,18]] (4,1) windows with the reference's guessing-decoder shape D4 / S20 (`Sliding Window GDG.ipynb` cell 8)
  config 5  SHYPS r=3 circuit-level p=0.001, 12 rounds, (3,1), osd_window windows (stim-free DEM, shyps.py)
  bp4       [[144,12,12]] depolarizing code-capacity noise, bp4_osd (Misc.ipynb cell 2 setting)
  order10   configs[1] with the notebooks' default OSD-CS order 10

One JSON line per configuration: windows (or decodes) per second from HIP-event kernel time."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder, bp4_osd
from slidingwindowdecoder_amd.windows import sample_dem
from slidingwindowdecoder_amd.codes import bb_code

which = sys.argv[1:] or ["3", "3small", "3ens", "3mt", "4", "4gdg", "5", "5w12", "bp4", "bp4shyps", "order10"]


def run_pipeline(name, plan, shots, reps, **kw):
    dec = SlidingWindowDecoder(plan, **kw)
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=int(os.environ.get("SWD_CFG_SEED", "7")))
    d = torch.from_numpy(np.ascontiguousarray(det)).cuda()
    stats = torch.empty((shots, dec.W, 8), dtype=torch.int32, device="cuda")
    dec.decode_device(d, stats=stats); torch.cuda.synchronize()
    dec.set_timing(True)
    for _ in range(reps):
        dec.decode_device(d, stats=stats)
    torch.cuda.synchronize()
    ms, n = dec.get_timing()
    dec.check_status()  # raises on a scheduling fault
    st = stats.cpu().numpy()
    conv = ((st[..., 0] & 0x100) != 0).mean()
    print(json.dumps({"config": name, "shots": shots, "windows_per_shot": dec.W, "ms_per_launch": ms / n,
                      "windows_per_s": shots * dec.W / (ms / n / 1e3), "threads": dec.threads, "lds_bytes": dec.lds_bytes,
                      "converged_fraction": float(conv), "exit_classes": np.bincount((st[..., 0] & 0xFF).ravel(), minlength=6).tolist()}), flush=True)


if "3" in which:
    run_pipeline("configs[2]: [[144,12,12]] p=0.003 (3,1) bpgdg_decoder(max_iter=8, T=6, R=25, D=3, S=10)", bench.build_problem(), int(os.environ.get("SWD_GDG_SHOTS", "16384")), 2,
                 decoder="bpgdg_decoder", max_iter=8, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
                 max_tree_branch_step=10, max_side_branch_step=10)
GDG_KW = dict(decoder="bpgdg_decoder", max_iter=8, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
              max_tree_branch_step=10, max_side_branch_step=10)
if "3small" in which:  # small batch: side branches of a shot's decimation tree run as work items on many workgroups
    run_pipeline("configs[2] at 2048 shots per launch (parallel tree search)", bench.build_problem(), 2048, 3, **GDG_KW)
if "3ens" in which:    # the 64-hypothesis ensemble (multi_thread semantics, no parity target)
    run_pipeline("configs[2], hypotheses=64 ensemble (D=5, S=6, every leaf scored), 2048 shots per launch", bench.build_problem(), 2048, 3,
                 **dict(GDG_KW, hypotheses=64))
if "3mt" in which:   # the reference's threaded ensemble (multi_thread=True: main + 7 tree + 7 side threads), one workgroup per (shot, window)
    run_pipeline("configs[2], bpgdg_decoder(multi_thread=True) = the reference's threaded ensemble (D=3, S=10: 15 thread bodies per window), 4096 shots per launch",
                 bench.build_problem(), 4096, 3, **dict(GDG_KW, multi_thread=True))
    run_pipeline("configs[2], threaded ensemble with D=5, S=6 (31 tree threads with two leaves each + main + 1 side: 64 hypotheses), 4096 shots per launch",
                 bench.build_problem(), 4096, 3, **dict(GDG_KW, multi_thread=True, max_tree_depth=5, max_side_depth=6))
if "4" in which:
    run_pipeline("configs[3]: [[288,12,18]] p=0.003 (4,1) osd_window(pre=8, post=200, osd_cs 0)", bench.build_problem(N=288, W=4, F=1), 4096, 2,
                 **dict(bench.DECODER_KW, osd_order=0))
if "4gdg" in which:  # the reference's [[288,12,18]] guessing-decoder run (`Sliding Window GDG.ipynb` cell 8 / cell 4: (4,1) windows, max_iter 16,
    # max_step 60, D4 / S20, branch steps 40), single-thread gdg() and the 32-thread ensemble, at the notebook's p = 0.005 / 6 rounds and at p = 0.003 / 12 rounds
    G288 = dict(decoder="bpgdg_decoder", max_iter=16, max_iter_per_step=6, max_step=60, max_tree_depth=4, max_side_depth=20,
                max_tree_branch_step=40, max_side_branch_step=40)
    for p288, rounds in ((0.005, 6), (0.003, 12)):
        plan288 = bench.build_problem(N=288, p=p288, rounds=rounds, W=4, F=1)
        run_pipeline(f"[[288,12,18]] p={p288} {rounds} rounds (4,1) bpgdg_decoder(max_iter=16, R=60, D=4, S=20, branch steps 40), gdg()", plan288, 2048, 2, **G288)
        run_pipeline(f"[[288,12,18]] p={p288} {rounds} rounds (4,1) bpgdg_decoder(multi_thread=True, max_iter=16, R=60, D=4, S=20): 32 threads", plan288, 2048, 2,
                     **dict(G288, multi_thread=True))
if "5" in which:
    from slidingwindowdecoder_amd import shyps
    from slidingwindowdecoder_amd.windows import plan_windows
    dem = shyps.shyps_dem(3, 0.001, 12)
    run_pipeline("configs[4] circuit: SHYPS r=3 p=0.001, 12 rounds, (3,1) windows 63x476, osd_window(pre=8, post=200, osd_cs 0) "
                 "(binary BP+OSD like SHYPS.ipynb; BP4 has no circuit-level reference)",
                 plan_windows(dem.chk, dem.obs, dem.priors, 21, 3, 1, method=1), 8192, 3, **dict(bench.DECODER_KW, osd_order=0))
if "5w12" in which:
    from slidingwindowdecoder_amd import shyps
    from slidingwindowdecoder_amd.windows import plan_windows
    dem = shyps.shyps_dem(3, 0.001, 14)
    run_pipeline("configs[4] circuit with twelve-round windows: SHYPS r=3 p=0.001, 14 rounds, (12,1) windows 252x2240, osd_window(pre=8, post=200, osd_cs 0)",
                 plan_windows(dem.chk, dem.obs, dem.priors, 21, 12, 1, method=1), 4096, 3, **dict(bench.DECODER_KW, osd_order=0))
if "order10" in which:
    run_pipeline("configs[1] with osd_cs order 10", bench.build_problem(), 4096, 3, **dict(bench.DECODER_KW, osd_order=10))
if "bp4" in which:
    code, _, _ = bb_code(144)
    n = code.hx.shape[1]
    p = 0.02
    pr = np.full(n, p / 3)
    dec = bp4_osd(code.hx, code.hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100,
                  ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10)
    rng = np.random.default_rng(5)
    B = 16384
    pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])  # 0 I, 1 X, 2 Y, 3 Z
    ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
    sx = (ez @ code.hx.T % 2).astype(np.uint8); sz = (ex @ code.hz.T % 2).astype(np.uint8)
    dec.decode_batch(sx[:256], sz[:256])
    t0 = time.perf_counter(); out = dec.decode_batch(sx, sz); dt = time.perf_counter() - t0
    dec.decode_batch(sx, sz, details=False)
    t0 = time.perf_counter(); out2 = dec.decode_batch(sx, sz, details=False); dt2 = time.perf_counter() - t0
    assert np.array_equal(out, out2)
    print(json.dumps({"config": "bp4_osd [[144,12,12]] depolarizing p=0.02, max_iter=100, osd_cs 10 (host buffers, PCIe included)",
                      "decodes": B, "decodes_per_s": B / dt, "decodes_per_s_decisions_only": B / dt2,
                      "converged_fraction": float(((dec.last_status & 0x100) != 0).mean())}), flush=True)

if "bp4shyps" in which:  # BASELINE config 5's decoder on config 5's code (code capacity: the setting the reference can run BP4 in)
    from slidingwindowdecoder_amd import shyps
    SX, SZ = shyps.shyps_stabilizers(3)
    n = SX.shape[1]
    p = 0.02
    pr = np.full(n, p / 3)
    dec = bp4_osd(SX, SZ, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=32, ms_scaling_factor=0.625,
                  osd_method="osd_cs", osd_order=10)
    rng = np.random.default_rng(6)
    B = 65536
    pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])
    ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
    sx = (ez @ SX.T % 2).astype(np.uint8); sz = (ex @ SZ.T % 2).astype(np.uint8)
    dec.decode_batch(sx[:256], sz[:256])
    t0 = time.perf_counter(); out = dec.decode_batch(sx, sz); dt = time.perf_counter() - t0
    dec.decode_batch(sx, sz, details=False)
    t0 = time.perf_counter(); out2 = dec.decode_batch(sx, sz, details=False); dt2 = time.perf_counter() - t0
    assert np.array_equal(out, out2)
    print(json.dumps({"config": "configs[4] decoder: bp4_osd on the SHYPS r=3 stabiliser matrices (21x49 each), depolarizing p=0.02, max_iter=32, osd_cs 10 "
                                "(host buffers, PCIe included)", "decodes": B, "decodes_per_s": B / dt, "decodes_per_s_decisions_only": B / dt2,
                      "converged_fraction": float(((dec.last_status & 0x100) != 0).mean())}), flush=True)
if "w2" in which:  # occupancy experiment (docs/history/DESIGN_rounds_1-5.md section 6): (2,1) windows of the [[144,12,12]] circuit need 34 KB of LDS
    run_pipeline("[[144,12,12]] p=0.003 (2,1) osd_window(pre=8, post=200, osd_cs 0)", bench.build_problem(W=2), 4096, 5, **bench.DECODER_KW)
