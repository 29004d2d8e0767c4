"""GPU: the long-runner lane (lpvmpc_lane_create / lpvmpc_lane_attach, include/lpvmpc.h).  Step streams masked to the CUs the
lane did not reserve, tail streams masked to the reserved ones; a deferred call promotes the instances that are still unsolved
after `promote_after` iterations into the handle's ring and a drain launch of the whole-CU tail kernel finishes them on the
reserved CUs while the step streams go on.  Nothing about the results may change against the plain launch: status, iteration
count and polish flag equal, solutions to the tail kernel's round-off (tests/test_gpu_deferral.py: 1e-7 polished, 1e-6 un-polished),
instances that were never parked bit-identical; the instances beyond 1000 iterations are compared with the CPU oracle."""
import numpy as np
import pytest

from test_gpu_deferral import _close_to, _dev_call, _host

pytestmark = pytest.mark.gpu


class _Raw:
    def __init__(self, p):
        self.cuda_stream = p


def _plain(torch, w, B, planner=False):
    from lpvmpc import workloads
    plain = workloads.make_solver(w); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, planner); torch.cuda.synchronize()
    ref = _host(o); plain.close()
    return ref


@pytest.mark.parametrize("promote_after,budget,remaining", [(200, 100, 0), (100, 100, 0), (100, -1, 0), (300, 100, 0), (100, 100, 400), (200, 100, 1000)])
def test_lane_finishes_promoted_stragglers(promote_after, budget, remaining):
    import torch
    import lpvmpc
    from lpvmpc import workloads
    B = 1024
    lane = lpvmpc.Lane(device=0, reserved_cus=8, step_streams=2, ring_entries=16)
    for seed in (3, 16, 19):                      # 16 / 19 hold the instances that run to OSQP's max_iter
        w = workloads.controller_batch(B, N=20, seed=seed)
        ref = _plain(torch, w, B)
        assert ref["iters"].max() >= 2400
        eng = workloads.make_solver(w); eng.reserve(B)
        promoted0 = lane.promoted_count()
        eng.set_option("defer_after", 100); eng.set_option("defer_budget", budget)
        eng.attach_lane(lane, promote_after=promote_after, promote_remaining=remaining)      # (remaining > 0: only what the residuals predict to be long)
        st = _Raw(lane.step_streams[seed % 2])
        keep = [_dev_call(torch, eng, w, B, False, stream=st) for _ in range(3)]          # three calls in flight on one step stream
        eng.join(st.cuda_stream); torch.cuda.synchronize()
        assert eng.lane_drain_count() >= 2 and lane.promoted_count() - promoted0 >= (2 if remaining < 1000 else 1)             # (each call holds an instance beyond 2400 iterations; the last call's may still be below promote_after at the join)
        for _, o in keep:
            got = _host(o)
            assert not np.any(got["status"] == -11)
            _close_to(got, ref)
            early = ref["iters"] <= 100
            assert np.array_equal(got["uPred"][early], ref["uPred"][early], equal_nan=True)
        eng.close()
    lane.close()


def test_lane_small_ring_overflows_into_the_pool():
    """A one-entry ring: whatever finds it taken stays with the bounded passes and the join's closing pass; results unchanged."""
    import torch
    import lpvmpc
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=3)
    ref = _plain(torch, w, B)
    lane = lpvmpc.Lane(device=0, reserved_cus=8, step_streams=1, ring_entries=1)
    eng = workloads.make_solver(w); eng.reserve(B)
    eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100)
    eng.attach_lane(lane, promote_after=100)
    st = _Raw(lane.step_streams[0])
    keep = [_dev_call(torch, eng, w, B, False, stream=st) for _ in range(2)]
    eng.join(st.cuda_stream); torch.cuda.synchronize()
    assert 1 <= lane.promoted_count() < 2 * int(np.sum(ref["iters"] > 100))
    for _, o in keep:
        _close_to(_host(o), ref)
    # detached again: the handle behaves as before
    eng.attach_lane(None)
    _, o = _dev_call(torch, eng, w, B, False)
    eng.join(0); torch.cuda.synchronize()
    _close_to(_host(o), ref)
    eng.close(); lane.close()


def test_lane_shared_by_controller_and_planner_handles():
    """Two handles (controller and planner, N = 20) on one lane, their calls interleaved on the two step streams."""
    import torch
    import lpvmpc
    from lpvmpc import workloads
    B = 512
    wc = workloads.controller_batch(B, N=20, seed=16)
    wp = workloads.planner_batch(B, N=20, seed=2)
    refc, refp = _plain(torch, wc, B), _plain(torch, wp, B, True)
    lane = lpvmpc.Lane(device=0, reserved_cus=8, step_streams=2, ring_entries=32)
    ec = workloads.make_solver(wc); ec.reserve(B); ep = workloads.make_solver(wp); ep.reserve(B)
    for e in (ec, ep):
        e.set_option("defer_after", 100); e.set_option("defer_budget", 100); e.attach_lane(lane, promote_after=300)
    s0, s1 = _Raw(lane.step_streams[0]), _Raw(lane.step_streams[1])
    kc = [_dev_call(torch, ec, wc, B, False, stream=s0) for _ in range(2)]
    kp = [_dev_call(torch, ep, wp, B, True, stream=s1) for _ in range(2)]
    ec.join(s0.cuda_stream); ep.join(s1.cuda_stream); torch.cuda.synchronize()
    assert lane.promoted_count() >= 16
    for _, o in kc:
        _close_to(_host(o), refc)
    for _, o in kp:
        _close_to(_host(o), refp, tol_polished=1e-7, tol_iterate=1e-5)
    ec.close(); ep.close(); lane.close()


def test_lane_arguments():
    import lpvmpc
    from lpvmpc import workloads
    with pytest.raises(lpvmpc.LpvMpcError):
        lpvmpc.Lane(device=0, reserved_cus=4)                 # not a multiple of 8: some XCC would be left without a reserved CU
    with pytest.raises(lpvmpc.LpvMpcError):
        lpvmpc.Lane(device=0, reserved_cus=8, ring_entries=0)
    lane = lpvmpc.Lane(device=0, reserved_cus=16, step_streams=0)
    assert lane.step_streams == []
    w = workloads.planner_batch(8, N=30, seed=1)
    eng = workloads.make_solver(w)
    with pytest.raises(lpvmpc.LpvMpcError):
        eng.attach_lane(lane)                                 # no tail kernel at N = 30
    eng.close(); lane.close()
