#!/usr/bin/env python3
"""Debug helper: one dumped window (tests/fuzz_pipeline.py, SWD_FUZZ_DUMP) through the device and the oracle under parameter variations."""
import json, os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
f = np.load(sys.argv[1])
mat = sp.csr_matrix((np.ones(len(f["indices"]), np.uint8), f["indices"], f["indptr"]), shape=tuple(f["shape"]))
prior, synd, kw0 = f["prior"], f["synd"], json.loads(str(f["kw"]))
print("shape", mat.shape, "kw", kw0, "synd weight", int(synd.sum()))
gpu = len(sys.argv) > 2
if gpu:
    from slidingwindowdecoder_amd import bpgdg_decoder
def run(tag, **over):
    kw = dict(kw0); kw.update(over)
    o = O.bpgdg_decoder(mat, channel_probs=prior, **kw)
    e_o = np.asarray(o.decode(synd))
    line = f"{tag:28s} oracle conv {o.converge} pm {o.min_pm:.6f} w {int(e_o.sum())}"
    if kw.get("multi_thread") and o._res.exit_class != 0:
        info = o.ensemble_info(); line += f" pms {[round(x, 3) for x in info[0].tolist()]} win {info[1]} blocks {o.ensemble_blocks()[0]}"
    if gpu:
        dv = bpgdg_decoder(mat, channel_probs=prior, **kw)
        e_d = np.asarray(dv.decode_batch(synd[None, :]))[0]
        line += f" | device equal {np.array_equal(e_d, e_o)} pm {dv.last_min_pm[0]:.6f} w {int(e_d.sum())} stats {dv.last_stats[0].tolist()} diff at {np.flatnonzero(e_d != e_o).tolist()}"
    print(line)
run("as dumped")
run("no sides", max_side_depth=0)
for k in range(1, 9): run(f"no sides max_step {k}", max_side_depth=0, max_step=k)
run("low_error_mode", low_error_mode=True)
run("single thread", multi_thread=False)
run("new_n = n", new_n=mat.shape[1])
run("iter per step 4", max_iter_per_step=4)
run("iter per step 8", max_iter_per_step=8)
run("max_iter 4", max_iter=4)
run("max_iter 8", max_iter=8)
