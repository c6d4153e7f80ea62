"""world_size-2 run of the shot-sharding path on CPU (gloo): every rank decodes its contiguous
shard (here with the CPU oracle standing in for the device pipeline -- tests may do that), one
all_gather of the per-shot decisions, and the gathered result equals the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slidingwindowdecoder_amd.distributed import decode_sharded, shard_bounds
from tests import fixtures as fx


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _oracle_decisions(det):
    """per-shot [obs-flip mask, flagged] computed with the oracle through the host window loop"""
    import scipy.sparse as sp
    from oracle import oracle as O
    from slidingwindowdecoder_amd.windows import sliding_window_decode_host
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "osd0_params")
    total, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    pred = (sp.csr_matrix(total) @ plan.obs.T.astype(np.int32)).toarray() % 2
    mask = (pred.astype(np.int64) << np.arange(pred.shape[1])).sum(axis=1)
    resid = (det + (sp.csr_matrix(total) @ plan.chk.T.astype(np.int32)).toarray()) % 2
    return np.stack([mask, resid.any(axis=1).astype(np.int64)], axis=1).astype(np.int32)


def _worker(rank, world, port, det, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = decode_sharded(det, _oracle_decisions)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    det = fx.unpack(f["det"], 936)[:21]  # odd count: uneven shards
    want = _oracle_decisions(det)
    # the recorded reference run: logical flags of these shots
    obs = fx.unpack(f["obs_data"], 12)[:21]
    obs_mask = (obs.astype(np.int64) << np.arange(12)).sum(axis=1)
    assert np.array_equal(((want[:, 0] != obs_mask) | (want[:, 1] != 0)).astype(np.uint8), f["osd0_logical"][:21])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, det, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
