#!/usr/bin/env python3
"""Where do the non-converging bp4_osd decodes of a batch sit in the heaviest-syndrome-first start order?"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import bp4_osd
from slidingwindowdecoder_amd.codes import bb_code
code, _, _ = bb_code(144)
n, p = 144, 0.02
pr = np.full(n, p / 3)
dec = bp4_osd(code.hx, code.hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10)
rng = np.random.default_rng([20240318, 0, 0])
B = 65536
pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])
ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
sx = np.ascontiguousarray((ez @ code.hx.T % 2).astype(np.uint8)); sz = np.ascontiguousarray((ex @ code.hz.T % 2).astype(np.uint8))
dev = torch.device("cuda", 0)
tx, tz = torch.from_numpy(sx).to(dev), torch.from_numpy(sz).to(dev)
out = torch.empty((B, 2, n), dtype=torch.uint8, device=dev); st = torch.empty((B, 8), dtype=torch.int32, device=dev)
dec.decode_batch_device(tx, tz, out=out, stats=st); torch.cuda.synchronize()
its = st.cpu().numpy()[:, 1]
w = sx.sum(1) + sz.sum(1)
order = np.argsort(-w, kind="stable")
rank = np.empty(B, int); rank[order] = np.arange(B)
heavy = np.flatnonzero(its >= 100)
print(json.dumps({"heavy": len(heavy), "ranks_of_heavy_in_start_order_fraction": sorted((rank[heavy] / B).round(3).tolist()),
                  "weights_of_heavy": sorted(w[heavy].tolist()), "weight_percentiles_all": np.percentile(w, [50, 90, 99, 99.9]).tolist(),
                  "iters_hist": np.bincount(np.minimum(its, 12)).tolist()}))
