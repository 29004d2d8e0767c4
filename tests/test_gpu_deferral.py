"""GPU: straggler deferral (lpvmpc_set_option "defer_after" / "defer_budget").  A deferred call parks the instances that are
still unsolved after K iterations; resume passes of the same kernel continue them, a budget of iterations at a time, and
lpvmpc_join finishes what is left.  Nothing about the results may change: status, iteration count, polish flag, residuals and
every word of xPred / uPred are BIT-IDENTICAL to the plain call (a restored instance re-factors K from its saved state, a pure
function of it) -- as long as the same kernel continues it ("defer_tail" 0).  By default the pass that runs the parked instances to
completion is the whole-CU tail kernel, which applies K^-1 as a dense matrix: same statuses, iteration counts and polish flags,
solutions equal to round-off (the tests at the end; tests/diagnostics/tail_sweep.py: 196 608 instances)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev_call(torch, eng, w, B, planner, stream=None):
    dev = torch.device("cuda", 0)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    N, nx = w["N"], (5 if planner else 6)
    ins = dict(x0=t(w["x0"]), u_prev=t(w["u_prev"]), vel=t(w["vel_ref"]), curv=t(w["curv_s"]), u_old=t(w["u_old"]), mey=t(w["max_ey"]))
    o = dict(xPred=torch.full((B, N + 1, nx), -7.0, dtype=torch.float64, device=dev), uPred=torch.full((B, N, 2), -7.0, dtype=torch.float64, device=dev),
             status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev),
             resid=torch.zeros((B, 4), dtype=torch.float64, device=dev), polish=torch.zeros(B, dtype=torch.int32, device=dev))
    eng.solve_dev(B, ins["x0"], ins["u_prev"], ins["vel"], ins["curv"], ins["u_old"], ins["mey"], o["xPred"], o["uPred"], o["status"], o["iters"],
                  o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"], stream=0 if stream is None else stream.cuda_stream)
    return ins, o


def _host(o):
    return {k: v.cpu().numpy() for k, v in o.items()}


def _same(a, b):
    for k in ("status", "iters", "polish"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("xPred", "uPred", "resid"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("K,budget", [(25, 50), (100, 200), (100, 0)])
def test_deferred_controller_batch_is_bit_identical(K, budget):
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=3)            # contains instances with 100 .. 2400 iterations
    plain = workloads.make_solver(w); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref = _host(o); plain.close()
    assert ref["iters"].max() >= 1000 and np.sum(ref["iters"] > K) >= 4
    eng = workloads.make_solver(w); eng.reserve(B)
    eng.set_option("defer_after", K); eng.set_option("defer_budget", budget); eng.set_option("defer_tail", 0)
    keep, o = _dev_call(torch, eng, w, B, False)
    torch.cuda.synchronize()
    if budget:                                # one bounded pass cannot have finished the 2400-iteration instance
        assert np.any(_host(o)["status"] == -11)
    eng.join(0); torch.cuda.synchronize()
    _same(_host(o), ref)
    # several calls in flight on one stream (more than the handle has pool segments) with different inputs each
    outs = []
    st = torch.cuda.Stream()
    for j in range(20):
        wj = workloads.controller_batch(256, N=20, seed=40 + j)
        outs.append((wj, _dev_call(torch, eng, wj, 256, False, stream=st)))
    eng.join(st.cuda_stream); torch.cuda.synchronize()
    chk = workloads.make_solver(w)
    for wj, (_, oj) in outs[::5]:
        r = chk.solve(wj["x0"], wj["u_prev"], wj["vel_ref"], wj["curv_s"], wj["u_old"], None, wj["cf_new"], wj["lap"])
        h = _host(oj)
        assert not np.any(h["status"] == -11)
        assert np.array_equal(h["iters"], r["iters"]) and np.array_equal(h["uPred"], r["uPred"], equal_nan=True)
    chk.close(); eng.close()


def test_deferral_pool_overflow_and_pending_status():
    """Pools too small for the stragglers: the surplus instances simply finish inside the launch that holds them; results
    unchanged."""
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=3)
    plain = workloads.make_solver(w)
    ref = plain.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]); plain.close()
    n_over = int(np.sum(ref["iters"] > 50))
    assert n_over > 8
    eng = workloads.make_solver(w); eng.reserve(B)
    eng.set_option("defer_pool", 4); eng.set_option("defer_after", 50); eng.set_option("defer_tail", 0)
    _, o = _dev_call(torch, eng, w, B, False)
    eng.join(0); torch.cuda.synchronize()
    h = _host(o)
    assert np.array_equal(h["iters"], ref["iters"]) and np.array_equal(h["status"], ref["status"])
    assert np.array_equal(h["uPred"], ref["uPred"], equal_nan=True) and np.array_equal(h["xPred"], ref["xPred"], equal_nan=True)
    eng.close()


def test_age_ordered_admission_parks_the_long_runners_of_an_overflowing_batch():
    """Round 6 (lpvmpc.h, "defer_pool"): with defer_after = 50 a seventh of this controller batch (154 instances) asks for a pool entry
    and the default pool (128 = B / 8) overflows.  Admission is by age class -- instances below 2 K iterations may take three quarters
    of the pool, those between 2 K and 4 K an eighth of their own, older ones any free entry -- so the batch's long runners (seed 16
    holds a 3900-iteration instance) are parked by their check at 4 K = 200 at the latest instead of staying resident in the main
    launch to their end; lpvmpc_defer_stats counts the parked instances and the refused requests; the results are those of the plain
    call, bit for bit ("defer_tail" 0)."""
    import torch
    from lpvmpc import workloads
    B, K = 1024, 50
    w = workloads.controller_batch(B, N=20, seed=16)
    plain = workloads.make_solver(w)
    ref = plain.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]); plain.close()
    assert ref["iters"].max() >= 3000 and int(np.sum(ref["iters"] > K)) > 128             # a many-thousand-iteration instance, and more candidates (154) than the pool holds
    eng = workloads.make_solver(w); eng.reserve(B)
    eng.set_option("defer_after", K); eng.set_option("defer_budget", -1); eng.set_option("defer_tail", 0)      # (no pass behind the call: what the main launch parked stays parked)
    assert eng.defer_stats() == (0, 0)
    _, o = _dev_call(torch, eng, w, B, False)
    torch.cuda.synchronize()
    h = _host(o)
    pending = h["status"] == -11
    parked, refused = eng.defer_stats()
    assert parked == int(pending.sum()) and 96 <= parked <= 128 and refused > 0, (parked, refused, int(pending.sum()))
    old = ref["iters"] > 4 * K                                    # the instances that pass 4 K iterations: every one was parked, none later than its check at 4 K
    assert old.sum() >= 1 and np.all(pending[old]) and np.all(h["iters"][old] <= 4 * K), (int(old.sum()), h["iters"][old], h["status"][old])
    assert np.all(h["iters"][pending] >= K)
    eng.join(0); torch.cuda.synchronize()
    h = _host(o)
    assert np.array_equal(h["iters"], ref["iters"]) and np.array_equal(h["status"], ref["status"]) and np.array_equal(h["polish"], ref["polish"])
    assert np.array_equal(h["uPred"], ref["uPred"], equal_nan=True) and np.array_equal(h["xPred"], ref["xPred"], equal_nan=True)
    assert eng.defer_stats()[0] == parked                         # the closing pass runs to completion: nothing is parked again
    eng.close()


def test_deferred_planner_batch_is_bit_identical():
    """The DPP two-wavefront planner kernel (N = 30) and the run-time-horizon kernel go through the same park / restore."""
    import torch
    from lpvmpc import workloads
    B = 256
    w = workloads.planner_batch(B, N=30, seed=1)
    for variant in (0, 1):
        plain = workloads.make_solver(w); plain.set_option("kernel_variant", variant); plain.reserve(B)
        _, o = _dev_call(torch, plain, w, B, True); torch.cuda.synchronize(); ref = _host(o); plain.close()
        eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant); eng.reserve(B)
        eng.set_option("defer_after", 100)
        _, o = _dev_call(torch, eng, w, B, True)
        eng.join(0); torch.cuda.synchronize()
        got = _host(o)
        assert np.array_equal(got["status"], ref["status"]), (variant, np.nonzero(got["status"] != ref["status"])[0][:10], got["status"][got["status"] != ref["status"]][:10], ref["status"][got["status"] != ref["status"]][:10], ref["iters"][got["status"] != ref["status"]][:10])
        _same(got, ref)
        eng.close()


def test_deferral_with_warm_start_bad_inputs_and_housekeeping():
    """Deferral next to the other options: the opt-in warm start (its state buffer travels with the parked instance), non-finite
    inputs (those instances never iterate, nothing to park), join without a call, switching the option off (implies a join),
    destroying a handle that still holds parked instances."""
    import torch
    from lpvmpc import workloads
    B = 512
    w = workloads.controller_batch(B, N=20, seed=3)
    w["x0"][7, 2] = np.nan                                    # one bad instance
    runs = {}
    for defer in (0, 50):
        eng = workloads.make_solver(w); eng.reserve(B)
        eng.set_option("warm_start", 2); eng.set_option("defer_budget", 75); eng.set_option("defer_tail", 0)
        eng.join(0)                                           # nothing to join yet
        eng.set_option("defer_after", defer)
        outs = []
        for _ in range(3):                                    # cold solve, then two warm-started ones
            _, o = _dev_call(torch, eng, w, B, False)
            eng.join(0); torch.cuda.synchronize()
            outs.append(_host(o))
        runs[defer] = outs
        if defer:
            _, o = _dev_call(torch, eng, w, B, False)         # leave instances parked ...
            eng.set_option("defer_after", 0)                  # ... switching off finishes them
            torch.cuda.synchronize()
            assert not np.any(_host(o)["status"] == -11)
            eng.set_option("defer_after", defer)
            _dev_call(torch, eng, w, B, False)                # parked again, never joined: close() must cope
        eng.close()
    for a, b in zip(runs[0], runs[50]):
        _same(a, b)
    assert runs[0][0]["status"][7] == -10 and np.all(np.isnan(runs[0][0]["uPred"][7]))
    assert runs[0][1]["iters"].sum() < runs[0][0]["iters"].sum()          # the warm start did take effect


def test_host_array_call_with_deferral_returns_finished_results():
    """lpvmpc_solve_batch (host arrays, synchronous) on a handle with straggler deferral: the call joins before it copies the
    outputs back, so no instance comes back LPVMPC_PENDING and every word equals the plain call's."""
    from lpvmpc import workloads
    w = workloads.controller_batch(1024, N=20, seed=16)          # seed 16 holds an instance that runs to max_iter
    plain = workloads.make_solver(w)
    ref = plain.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    plain.close()
    eng = workloads.make_solver(w)
    eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100); eng.set_option("defer_tail", 0)
    got = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    eng.set_option("defer_tail", 1)
    got_tail = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    eng.close()
    assert ref["iters"].max() > 1000 and not np.any(got["status"] == -11) and not np.any(got_tail["status"] == -11)
    for k in ("status", "iters", "polish", "xPred", "uPred", "resid"):
        assert np.array_equal(ref[k], got[k], equal_nan=True), k
    _close_to(got_tail, ref)


# ---- the whole-CU tail kernel (default for the passes that run to completion) ---------------------------------------------
def _close_to(got, ref, tol_polished=1e-7, tol_iterate=1e-6):
    """Same decisions, solutions equal to round-off: a polished solution is recomputed from the active set by two different
    factorisations of the 1/delta-weighted system (cond ~ 3e9: 2e-8 per solve; 1e-7 here, 1.4e-8 the largest seen over 196 608 instances), an ADMM iterate
    carries the K^-1 round-off of up to 4000 iterations (1e-6 relative to the largest entry; observed 1e-10)."""
    for k in ("status", "iters", "polish"):
        assert np.array_equal(got[k], ref[k]), (k, np.nonzero(got[k] != ref[k])[0][:8])
    has = np.isfinite(ref["uPred"]).all(axis=(1, 2))
    assert np.array_equal(has, np.isfinite(got["uPred"]).all(axis=(1, 2)))
    pol = ref["polish"] == 1
    for k in ("xPred", "uPred"):
        d = np.abs(got[k] - ref[k]).reshape(len(has), -1).max(axis=1)
        scale = np.maximum(1.0, np.abs(ref[k]).reshape(len(has), -1).max(axis=1))
        assert np.all(d[has & pol] <= tol_polished * scale[has & pol]), (k, d[has & pol].max())
        assert np.all(d[has & ~pol] <= tol_iterate * scale[has & ~pol]), (k, d[has & ~pol].max())


@pytest.mark.parametrize("seed", [3, 16, 19])
def test_tail_kernel_finishes_parked_stragglers(seed):
    """Deferred call + join with the tail kernel against the plain launch of the same batch: every status, iteration count and
    polish flag equal (seeds 16 / 19 hold the instances that run to OSQP's max_iter = 4000), solutions to round-off; instances that
    were never parked are bit-identical."""
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=seed)
    plain = workloads.make_solver(w); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref = _host(o); plain.close()
    assert ref["iters"].max() >= 2400
    for budget in (100, 0, -1):                    # -1: no pass behind the call, the join's tail pass takes everything
        eng = workloads.make_solver(w); eng.reserve(B)
        eng.set_option("defer_after", 100); eng.set_option("defer_budget", budget)
        _, o = _dev_call(torch, eng, w, B, False)
        if budget == -1:
            torch.cuda.synchronize()
            assert np.sum(_host(o)["status"] == -11) == np.sum(ref["iters"] > 100)       # everything unsolved at 100 is still parked
        eng.join(0); torch.cuda.synchronize()
        got = _host(o)
        _close_to(got, ref)
        early = ref["iters"] <= 100
        assert np.array_equal(got["uPred"][early], ref["uPred"][early], equal_nan=True)
        eng.close()


def test_tail_kernel_max_iter_instances_against_the_oracle():
    """The instances of seeds 16 / 19 that need more than 1000 iterations, finished by the tail kernel, against the CPU oracle
    (sparse LDL' of the full KKT system): equal status and iteration count, solutions within the stated tolerance."""
    import torch
    from lpvmpc import workloads
    from oracle import lpv_ref as L, osqp_ref as O
    p = dict(L.DEFAULT_PARAMS)
    n_max_iter = 0
    for seed in (16, 19):
        B = 1024
        w = workloads.controller_batch(B, N=20, seed=seed)
        eng = workloads.make_solver(w); eng.reserve(B)
        eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100)
        _, o = _dev_call(torch, eng, w, B, False)
        eng.join(0); torch.cuda.synchronize()
        got = _host(o); eng.close()
        slow = np.nonzero(got["iters"] > 1000)[0]
        assert len(slow) >= 1
        for b in slow:
            _, A, Bm = L.ctrl_lpv_prediction(p, w["dt"], w["N"], w["track"], w["x0"][b], w["u_prev"][b], w["vel_ref"][b], w["curv_s"][b], w["cf_new"], w["lap"])
            qp = L.ctrl_build_qp(w["Q"], w["R"], w["dR"], w["N"], A, Bm, w["x0"][b], w["u_old"][b], w["vel_ref"][b], p["max_vel"])
            r = O.solve_qp(qp.P, qp.q, qp.A, qp.l, qp.u)
            assert int(got["status"][b]) == r.info.status_val and int(got["iters"][b]) == r.info.iter, (seed, b, got["status"][b], got["iters"][b], r.info.status_val, r.info.iter)
            assert int(got["polish"][b]) == r.info.status_polish
            xP, uP, _ = L.unpack_solution(r.x, 6, 2, w["N"])
            if not np.all(np.isfinite(r.x)):                 # no solution (an infeasibility certificate): NaN on both sides
                assert np.all(np.isnan(got["uPred"][b])) and np.all(np.isnan(got["xPred"][b]))
                n_max_iter += int(r.info.iter >= 3900)
                continue
            tol = 1e-6 if r.info.status_polish == 1 else 2e-4
            assert np.max(np.abs(got["uPred"][b] - uP)) <= tol and np.max(np.abs(got["xPred"][b] - xP)) <= tol * max(1.0, np.max(np.abs(xP)))
            n_max_iter += int(r.info.iter >= 3900)
    assert n_max_iter >= 2          # seed 16: 3900 iterations, seed 19: OSQP's max_iter = 4000


@pytest.mark.parametrize("variant", [2, 3])
def test_tail_kernel_continues_entries_parked_by_the_other_compile_time_kernels(variant):
    """The one-wavefront (variant 2) and the DPP two-wavefront kernel (variant 3) park the same LDS image as the MFMA kernel: the
    tail kernel finishes their entries too (the run-time-horizon kernel, variant 1, keeps its own resume path: bit-identical)."""
    import torch
    from lpvmpc import workloads
    B = 512
    w = workloads.controller_batch(B, N=20, seed=3)
    plain = workloads.make_solver(w); plain.set_option("kernel_variant", variant); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref = _host(o); plain.close()
    assert np.sum(ref["iters"] > 100) >= 3
    eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant); eng.reserve(B)
    eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100)
    _, o = _dev_call(torch, eng, w, B, False)
    eng.join(0); torch.cuda.synchronize()
    _close_to(_host(o), ref)
    eng.set_option("kernel_variant", 1)                       # variant 1 parks another image: resumed by its own kernel, bit-identical
    plain = workloads.make_solver(w); plain.set_option("kernel_variant", 1); plain.reserve(B)
    _, o1 = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref1 = _host(o1); plain.close()
    _, o = _dev_call(torch, eng, w, B, False)
    eng.join(0); torch.cuda.synchronize()
    _same(_host(o), ref1)
    eng.close()


def test_tail_kernel_with_steering_delay_rows():
    """steeringDelay = 2 (CTRL:518-527): the pinned-steering rows are a seventh box row on the first two stages; the tail kernel
    carries them through its right-hand side, update and factorisation like the other kernels."""
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=3)
    rng = np.random.default_rng(11)
    w = dict(w, u_old=np.concatenate((w["u_old"], rng.normal(0, 0.08, (B, 2))), axis=1))      # [OldSteering[0], OldAccelera[0], OldSteering[1..2]]
    outs = {}
    for tail in (None, 1):
        eng = workloads.make_solver(w, steering_delay=2); eng.reserve(B)
        if tail is not None:
            eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100)
        _, o = _dev_call(torch, eng, w, B, False)
        eng.join(0); torch.cuda.synchronize()
        outs[tail] = _host(o); eng.close()
    assert np.sum(outs[None]["iters"] > 100) >= 4
    _close_to(outs[1], outs[None])
    # the pinned rows are honoured as OSQP honours them (polish regularises active rows with delta = 1e-6: violation = delta x multiplier)
    pol = outs[1]["polish"] == 1
    assert pol.sum() > 900 and np.median(np.abs(outs[1]["uPred"][pol, :2, 0] - w["u_old"][pol, 2:4])) < 1e-9


def test_deferred_call_on_the_null_stream_then_another_stream():
    """The null stream is a stream like any other: a deferred call on stream 0 followed -- without a synchronisation -- by a
    deferred call and the join on another stream is ordered by the handle's event hand-over (the pools are not touched by the
    second stream's launches while the first stream's resume pass still writes them)."""
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=3)
    plain = workloads.make_solver(w); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref = _host(o); plain.close()
    eng = workloads.make_solver(w); eng.reserve(B)
    eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100); eng.set_option("defer_tail", 0)
    st = torch.cuda.Stream()
    keep = []
    for rep in range(3):
        keep.append(_dev_call(torch, eng, w, B, False))                      # stream 0 (the null stream)
        keep.append(_dev_call(torch, eng, w, B, False, stream=st))           # straight on to another stream: no synchronisation
    eng.join(st.cuda_stream); torch.cuda.synchronize()
    for _, o in keep:
        _same(_host(o), ref)
    eng.close()


def test_tail_kernel_for_the_planner_at_n20():
    """The planner at N = 20 (configs[3]'s planner half: mean 640 iterations, a fifth at max_iter) has the same 168 unknowns:
    its parked instances are finished by the tail kernel too.  Against the plain launch: statuses, iteration counts and polish
    flags equal; solutions as close as two kernels get on this nearly-LP problem (polished 1e-7; an un-polished iterate after up to
    4000 iterations 1e-5, the tolerance of the kernel-variant cross-check)."""
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.planner_batch(B, N=20, seed=2)
    plain = workloads.make_solver(w); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, True); torch.cuda.synchronize(); ref = _host(o); plain.close()
    assert np.sum(ref["iters"] > 300) > 100 and ref["iters"].max() == 4000
    eng = workloads.make_solver(w); eng.reserve(B)
    eng.set_option("defer_after", 300); eng.set_option("defer_budget", -1); eng.set_option("defer_pool", 1024)
    _, o = _dev_call(torch, eng, w, B, True)
    eng.join(0); torch.cuda.synchronize()
    got = _host(o)
    ne = np.nonzero((got["status"] != ref["status"]) | (got["iters"] != ref["iters"]) | (got["polish"] != ref["polish"]))[0]
    print("planner N=20 tail: %d parked, %d decisions differ" % (int((ref["iters"] > 300).sum()), len(ne)), got["status"][ne][:8], ref["status"][ne][:8], got["iters"][ne][:8], ref["iters"][ne][:8])
    _close_to(got, ref, tol_polished=1e-7, tol_iterate=1e-5)
    eng.close()


@pytest.mark.parametrize("settings", [dict(check_termination=5), dict(check_termination=10, adaptive_rho_interval=50),
                                      dict(check_termination=25, max_iter=1010), dict(check_termination=7, max_iter=333, adaptive_rho_interval=21),
                                      dict(check_termination=25, adaptive_rho=0), dict(check_termination=100, adaptive_rho_interval=100)])
def test_tail_kernel_off_chain_checks_under_other_schedules(settings):
    """The tail kernel evaluates a termination check BESIDE the iterations (round 5): snapshot at the check iteration, verdict some
    iterations later, roll-back on a rho change.  Schedules other than OSQP's defaults stress that machinery: a check interval shorter
    than a check's flight (every pending verdict is then waited for before the next snapshot), rho updates that are not on every check,
    a max_iter that is not a check iteration (the closing evaluation), no rho adaptation at all.  Against the plain launch of the same
    settings: every status, iteration count and polish flag equal; solutions to the tail path's tolerance."""
    import torch
    from lpvmpc import workloads
    B = 1024
    w = workloads.controller_batch(B, N=20, seed=16)
    plain = workloads.make_solver(w, **settings); plain.reserve(B)
    _, o = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref = _host(o); plain.close()
    park = min(5 * settings["check_termination"], 200)
    assert np.sum(ref["iters"] > park) >= 3
    eng = workloads.make_solver(w, **settings); eng.reserve(B)
    eng.set_option("defer_after", park); eng.set_option("defer_budget", -1); eng.set_option("defer_pool", 1024)
    _, o = _dev_call(torch, eng, w, B, False)
    eng.join(0); torch.cuda.synchronize()
    got = _host(o)
    _close_to(got, ref, tol_polished=1e-7, tol_iterate=1e-6)
    eng.close()


def test_tail_parity_sweep_slice():
    """A slice of tests/diagnostics/tail_sweep.py inside the suite (four tracks x two seeds x lap 1 / lap 0 x 2048 controller
    instances = 32 768): the deferred call finished by the whole-CU tail kernel against the plain launch of the same batch --
    every status, iteration count and polish flag equal, never-parked instances bit-identical, parked ones to round-off -- and the
    instances beyond 1000 iterations against the CPU oracle (status and iteration count)."""
    from lpvmpc import workloads
    from oracle import osqp_ref as O
    n_parked = n_long = 0
    worst_pol = worst_it = 0.0
    for shape in ("oval", "L_shape", "3110", "Euge_Track"):
        for seed in (300, 301):
            for lap in (1, 0):
                B = 2048
                w = workloads.controller_batch(B, N=20, seed=seed, shape=shape)
                w["lap"] = lap
                plain = workloads.make_solver(w)
                ref = plain.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], lap); plain.close()
                eng = workloads.make_solver(w)
                eng.set_option("defer_after", 100); eng.set_option("defer_budget", 100)
                got = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], lap); eng.close()
                parked = ref["iters"] > 100
                n_parked += int(parked.sum())
                for k in ("status", "iters", "polish"):
                    assert np.array_equal(got[k], ref[k]), (shape, seed, lap, k, np.nonzero(got[k] != ref[k])[0][:8])
                fin = np.isfinite(ref["uPred"]).all(axis=(1, 2))
                assert np.array_equal(fin, np.isfinite(got["uPred"]).all(axis=(1, 2)))
                assert np.array_equal(got["uPred"][~parked], ref["uPred"][~parked], equal_nan=True)
                d = np.abs(got["uPred"] - ref["uPred"]).max(axis=(1, 2))
                pol = ref["polish"] == 1
                if (fin & pol & parked).any():
                    worst_pol = max(worst_pol, float(d[fin & pol & parked].max()))
                if (fin & ~pol & parked).any():
                    worst_it = max(worst_it, float(d[fin & ~pol & parked].max()))
                long_ = np.nonzero(ref["iters"] > 1000)[0]
                if len(long_):
                    sub = {k: (v[long_] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k != "track" else v) for k, v in w.items()}
                    orc = O.ctrl_tick_batch(sub, nthreads=16)
                    sane = orc["status"] != -10                      # (the oracle gives up on a singular factorisation: listed in DESIGN.md)
                    n_long += int(sane.sum())
                    assert np.array_equal(got["status"][long_][sane], orc["status"][sane]) and np.array_equal(got["iters"][long_][sane], orc["iters"][sane])
    assert n_parked >= 100 and n_long >= 5, (n_parked, n_long)
    assert worst_pol <= 1e-7 and worst_it <= 1e-6, (worst_pol, worst_it)
    print("tail sweep slice: %d parked, %d beyond 1000 iterations checked against the oracle, max |du| %.2e polished / %.2e un-polished" % (n_parked, n_long, worst_pol, worst_it))
