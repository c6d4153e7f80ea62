#!/usr/bin/env python3
"""Times the REFERENCE's own compiled osd_window (Cython, built by make_golden.ensure_reference in a scratch
directory) and this repository's CPU oracle (oracle/swd_oracle.c, the "port" of bench.py's cpu_baseline) on the
same recorded [[144,12,12]] p = 0.003 (3,1) sliding run -- the syndromes of tests/golden/bb144_circuit_p003_w3f1.npz,
OSD-CS order 10, one thread each -- and writes profiles/r06_cpu_reference_vs_port.json.  bench.py carries that figure
in `cpu_baseline` so that the GPU / CPU ratio is not read against the port alone.

Runs only in the build container (needs /root/reference); nothing under tests/ or bench.py imports it."""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
from tests import fixtures as fx  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main(reps=3):
    mg.ensure_reference()
    from src.osd_window import osd_window as ref_osd_window
    from oracle import oracle as O
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, "osd10_params")
    windows, traces = [], []
    for wi in range(11):
        mat, priors = fx.graph(f, f"win{wi}_")
        windows.append((mat, priors))
        traces.append(fx.Trace(f, f"osd10_win{wi}_", *mat.shape))
    decodes = sum(len(t) for t in traces)

    def run(make):
        decs = [make(mat, priors) for mat, priors in windows]
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            for d, tr in zip(decs, traces):
                for k in range(len(tr)):
                    out = d.decode(tr.synd[k])
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        assert (np.asarray(out, dtype=np.uint8) == traces[-1].out[-1]).all()
        return decodes / best

    ref = run(lambda mat, priors: ref_osd_window(mat, channel_probs=priors, **kw))
    port = run(lambda mat, priors: O.osd_window(mat, channel_probs=priors, **kw))
    rec = {"workload": "tests/golden/bb144_circuit_p003_w3f1.npz: the recorded (3,1) sliding run, 192 shots x 11 windows, "
                       "osd_window(pre=8, post=200, alpha=1.0, osd_cs order 10), one decode() call per window like osd.py:166-167",
           "decodes": decodes, "threads": 1, "cpu": cpu_model(), "best_of": reps,
           "reference_cython_windows_per_s_per_core": ref, "port_windows_per_s_per_core": port, "port_vs_reference": port / ref,
           "survey_probe": "SURVEY.md section 6 [probe]: 500-510 windows/s/core for the reference's osd_window in this container"}
    path = os.path.join(ROOT, "profiles", "r06_cpu_reference_vs_port.json")
    json.dump(rec, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
