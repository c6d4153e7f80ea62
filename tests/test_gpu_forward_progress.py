"""Forward progress of the persistent grids under reduced residency (round-5 verdict, weak item 8): the two lanes of a stream object
keep two launches in flight, each a persistent grid sized for the whole device, whose workgroups spin-wait for the predecessor
window of their shot.  The argument for progress -- a ticket's predecessor was drawn earlier, by a workgroup that is resident -- must
hold when a FOREIGN kernel owns half of the CUs for the whole time and most workgroups of both grids cannot become resident:
results equal the one-shot decode, no scheduling fault, and the foreign kernel really was running next to the launches."""
import time

import numpy as np
import pytest

from tests import fixtures as fx
from tests.test_gpu_pipeline import load_plan

pytestmark = pytest.mark.gpu


def test_two_lanes_progress_next_to_a_foreign_kernel_holding_half_the_cus():
    import torch
    from slidingwindowdecoder_amd import SlidingWindowDecoder, _lib
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "osd10_params")
    det = fx.unpack(f["det"], plan.chk.shape[0])
    want = fx.unpack(f["osd10_total"], plan.chk.shape[1])
    reps = 12                                           # 2304 shots per batch: more units than the (halved) device holds at once
    det_big, want_big = np.tile(det, (reps, 1)), np.tile(want, (reps, 1))
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    dec = SlidingWindowDecoder(plan, **kw)
    ncol = plan.chk.shape[1]
    d_t = torch.from_numpy(det_big).to(dev)
    outs = [dict(total=torch.zeros((len(det_big), ncol), dtype=torch.uint8, device=dev),
                 stats=torch.zeros((len(det_big), 11, 8), dtype=torch.int32, device=dev),
                 shot_result=torch.zeros((len(det_big), 2), dtype=torch.int32, device=dev)) for _ in range(2)]
    s = dec.stream(len(det_big))
    # undisturbed reference run (and warm-up)
    s.push_device(d_t, **outs[0])
    s.wait()
    assert np.array_equal(outs[0]["total"].cpu().numpy(), want_big)
    def four_launches(**kw):                            # both lanes, two launches in flight all the time
        t0 = time.perf_counter()
        for i in range(4):
            s.push_device(d_t, **kw, **outs[i % 2])
        s.wait()
        return time.perf_counter() - t0
    alone = min(four_launches() for _ in range(3))      # (the smallest of three: clocks and allocations have settled)
    dec.check_status()
    for o in outs:
        o["total"].zero_()
    # the foreign kernel: one 1024-thread workgroup with 150 KB of LDS per CU on half of the CUs for 1.5 s (no second such block fits a
    # CU, and what is left of the CU's 160 KB holds no pipeline workgroup: 53.6 KB each)
    # (The foreign stream may land on the hardware queue of a lane -- the runtime lets streams share its few queues -- and then the
    # lane's launches simply queue behind the foreign kernel: nothing ran side by side, nothing was tested.  Another stream, kept
    # beside the earlier ones, lands elsewhere: up to four attempts.)
    L = _lib.lib()
    streams = []
    for attempt in range(4):
        foreign = torch.cuda.Stream(device=dev)
        streams.append(foreign)
        done = torch.cuda.Event()
        torch.cuda.synchronize()
        assert L.swd_diag_occupy(0, cus // 2, 1024, 150 * 1024, 1_500_000, foreign.cuda_stream) == 0, _lib.last_error()
        done.record(foreign)
        time.sleep(0.05)                                   # the foreign grid is resident before the first launch
        crowded, overlapped = 1e9, True
        for _ in range(3):
            crowded = min(crowded, four_launches(after=False))
            overlapped = overlapped and not done.query()   # the launches finished while the foreign kernel was still running
        if overlapped:
            break
        print(f"attempt {attempt}: the launches ended after the foreign kernel (shared hardware queue?); once more on another stream")
    for k, o in enumerate(outs):
        got = o["total"].cpu().numpy()
        bad = np.flatnonzero((got != want_big).any(axis=1))
        assert bad.size == 0, f"lane {k}: {bad.size} shots differ next to the foreign kernel"
        assert ((o["stats"].cpu().numpy()[..., 0] & 0xFF) != 6).all(), "a window recorded a scheduling fault"
    dec.check_status()                                 # swd_pipeline_status == 0
    torch.cuda.synchronize()
    assert overlapped, f"the foreign kernel ended before the launches did (alone {alone:.3f} s, crowded {crowded:.3f} s): no overlap tested"
    assert crowded > 1.15 * alone, f"the launches were not slowed down by the foreign kernel (alone {alone:.3f} s, crowded {crowded:.3f} s)"
    print(f"four launches of {len(det_big)} shots (best of three): alone {alone * 1e3:.1f} ms, next to a foreign kernel on {cus // 2} CUs {crowded * 1e3:.1f} ms")
