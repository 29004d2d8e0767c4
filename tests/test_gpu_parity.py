"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(liblpvmpc.so via ctypes), against the CPU oracle and the golden vectors.

Tolerances (float64 path):
  * LPV evaluation / roll-out vs golden:   1e-11 relative to the largest magnitude of the compared array
    (device sin/cos and fused-multiply-add differ from numpy by a few ulp; the roll-out amplifies them).
  * QP solution vs the oracle's OSQP restatement on identical data: 1e-6 absolute on xPred/uPred when both
    polished (both then sit on the active-set optimum), 2e-4 otherwise (un-polished ADMM iterates at
    eps = 1e-3; the two implementations factor different-but-equivalent KKT forms);
    iteration counts and status codes must match exactly.
"""
import numpy as np
import pytest

from oracle import kkt_cert, lpv_ref as L, osqp_ref as O
from tests._golden import cases, load

pytestmark = pytest.mark.gpu

P = dict(L.DEFAULT_PARAMS)


@pytest.fixture(scope="module")
def lpvmpc():
    import lpvmpc as m
    return m


def relclose(a, b, tol):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape
    scale = max(1.0, float(np.max(np.abs(b))))
    err = float(np.max(np.abs(a - b)))
    assert err <= tol * scale, "max err %.3e > %.1e * %.3e" % (err, tol, scale)


def ctrl_engine(lpvmpc, c, track):
    return lpvmpc.BatchedSolver("controller", int(c["N"]), float(c["dt"]), c["Q"], c["R"], c["dR"], track=track)


def vfull(c):
    N = int(c["N"])
    v = np.asarray(c["vel_ref"], float).reshape(-1)
    return np.concatenate([v[:N], v[-1:]])


@pytest.mark.parametrize("name", ["ctrl_n10_cfg1", "ctrl_n20_oval"])
def test_controller_lpv_matches_golden(lpvmpc, name):
    tab = lpvmpc.Map("oval", 0.2).PointAndTangent
    for c in cases(name):
        eng = ctrl_engine(lpvmpc, c, tab)
        S, A, B = eng.lpv(c["x0"][None], c["u_prev"][None], vfull(c)[None], c["curv_ref"][None],
                          cf_new=float(c["cf_new"]), lap=int(c["lap"]))
        relclose(S[0], c["states"], 1e-11); relclose(A[0], c["A"], 1e-11); relclose(B[0], c["B"], 1e-11)
        eng.close()


@pytest.mark.parametrize("name", ["plan_n30_lshape", "plan_n40_lshape"])
def test_planner_lpv_matches_golden(lpvmpc, name):
    tab = lpvmpc.Map("L_shape", 0.2).PointAndTangent
    for c in cases(name):
        eng = lpvmpc.BatchedSolver("planner", int(c["N"]), float(c["dt"]), c["Q"], c["R"], c["dR"], L_cf=c["L_cf"], track=tab)
        S, A, B = eng.lpv(c["x0"][None], c["u_prev"][None], None, c["SS"][None])
        relclose(S[0], c["states"], 1e-10); relclose(A[0], c["A"], 1e-10); relclose(B[0], c["B"], 1e-10)
        eng.close()


def test_seed_mode_matches_golden(lpvmpc):
    g = load("seed_mode")
    eng = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, np.eye(6), np.eye(2), np.ones(2),
                               track=lpvmpc.Map("oval", 0.2).PointAndTangent)
    A, B = eng.estimate_abc(g["ctrl_xx"][None, :20], g["ctrl_uu"][None, :20, 0])
    relclose(A[0], g["ctrl_A"], 1e-12); relclose(B[0], g["ctrl_B"], 1e-12)
    eng.close()
    eng = lpvmpc.BatchedSolver("planner", 30, 0.05, np.eye(5), np.eye(2), np.ones(2), L_cf=np.zeros(5),
                               track=lpvmpc.Map("L_shape", 0.2).PointAndTangent)
    A, B = eng.estimate_abc(g["plan_xx"][None, :30], g["plan_uu"][None, :30])
    relclose(A[0], g["plan_A"], 1e-12); relclose(B[0], g["plan_B"], 1e-12)
    eng.close()


@pytest.mark.parametrize("name", ["ctrl_n10_cfg1", "ctrl_n20_oval"])
def test_controller_solve_matches_oracle_and_optimum(lpvmpc, name):
    tab = lpvmpc.Map("oval", 0.2).PointAndTangent
    for i, c in enumerate(cases(name)):
        N = int(c["N"])
        eng = ctrl_engine(lpvmpc, c, tab)
        out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], vfull(c)[None], c["old_u"][None])
        eng.close()
        assert int(out["status"][0]) == int(c["status_orc"]), (name, i)
        assert int(out["iters"][0]) == int(c["iter_orc"]), (name, i, out["iters"][0], c["iter_orc"])
        assert int(out["polish"][0]) == int(c["polish_orc"]), (name, i)
        xP, uP, _ = L.unpack_solution(c["x_orc"], 6, 2, N)
        tol = 1e-6 if int(c["polish_orc"]) == 1 else 2e-4
        assert np.max(np.abs(out["xPred"][0] - xP)) <= tol * max(1.0, np.max(np.abs(xP))), (name, i)
        assert np.max(np.abs(out["uPred"][0] - uP)) <= tol, (name, i)
        # independent check: KKT residuals of the returned primal point with the certified multipliers
        z = np.concatenate([out["xPred"][0].reshape(-1), out["uPred"][0].reshape(-1)])
        ok, info = kkt_cert.osqp_termination_ok(c["P"], c["q"], c["Aqp"], np.where(c["l"] < -1e29, -np.inf, c["l"]),
                                                c["u"], z, c["y_orc"])
        assert ok, info


@pytest.mark.parametrize("name", ["plan_n30_lshape", "plan_n40_lshape"])
def test_planner_solve_matches_oracle(lpvmpc, name):
    tab = lpvmpc.Map("L_shape", 0.2).PointAndTangent
    for i, c in enumerate(cases(name)):
        N = int(c["N"])
        eng = lpvmpc.BatchedSolver("planner", N, float(c["dt"]), c["Q"], c["R"], c["dR"], L_cf=c["L_cf"], track=tab)
        out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], None, np.zeros((1, 2)), max_ey=float(c["max_ey"]))
        eng.close()
        assert int(out["status"][0]) == int(c["status_orc"]), (name, i, out["status"][0], c["status_orc"], out["iters"][0], c["iter_orc"])
        assert int(out["iters"][0]) == int(c["iter_orc"]), (name, i, out["iters"][0], c["iter_orc"])
        if not np.all(np.isfinite(c["x_orc"])):
            assert np.all(np.isnan(out["xPred"][0])) and np.all(np.isnan(out["uPred"][0]))
            continue
        assert int(out["polish"][0]) == int(c["polish_orc"]), (name, i)
        xP, uP, _ = L.unpack_solution(c["x_orc"], 5, 2, N)
        tol = 1e-6 if int(c["polish_orc"]) == 1 else 2e-4
        assert np.max(np.abs(out["xPred"][0] - xP)) <= tol * max(1.0, np.max(np.abs(xP))), (name, i)
        assert np.max(np.abs(out["uPred"][0] - uP)) <= tol, (name, i)


def test_fused_batch_cfg2_against_oracle(lpvmpc):
    """BASELINE cfg 2 shape (controller N=20, random x0 on the oval), B = 256: every instance solved,
    a sample re-solved by the oracle from scratch (LPV evaluation + assembly + OSQP restatement)."""
    from lpvmpc import workloads
    B = 256
    w = workloads.controller_batch(B, N=20, seed=0)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    eng.close()
    assert np.all(out["status"] == 1), np.unique(out["status"], return_counts=True)
    for b in range(0, B, 8):
        S, A, Bm = L.ctrl_lpv_prediction(P, w["dt"], w["N"], w["track"], w["x0"][b], w["u_prev"][b], w["vel_ref"][b],
                                         w["curv_s"][b], w["cf_new"], w["lap"])
        qp = L.ctrl_build_qp(w["Q"], w["R"], w["dR"], w["N"], A, Bm, w["x0"][b], w["u_old"][b], w["vel_ref"][b], P["max_vel"])
        r = O.solve_qp(qp.P, qp.q, qp.A, qp.l, qp.u)
        xP, uP, _ = L.unpack_solution(r.x, 6, 2, w["N"])
        assert int(out["iters"][b]) == r.info.iter, (b, out["iters"][b], r.info.iter)
        tol = 1e-6 if (r.info.status_polish == 1 and out["polish"][b] == 1) else 2e-4
        assert np.max(np.abs(out["xPred"][b] - xP)) <= tol * max(1.0, np.max(np.abs(xP))), b
        assert np.max(np.abs(out["uPred"][b] - uP)) <= tol, b


def test_kernel_variants_agree(lpvmpc):
    """Four kernels run the same algorithm: variant 0 (default: compile-time horizon, two wavefronts per instance, two-sided
    elimination; controller N = 20 and planner N = 40: sweeps and factorisation on the matrix cores), variant 3 (the same with
    DPP sweeps: what the planner N = 30 kernel uses), variant 2 (compile-time horizon, one wavefront) and variant 1 (run-time
    horizon, factor tiles in LDS): identical statuses / polish flags, iteration counts equal (the factorisations differ in
    elimination order, i.e. in round-off only), solutions equal to 5e-6 (1e-8 when polished).  This cross-check is also the run-time guard behind the
    build's assembly scan (Makefile: a toolchain that mis-compiles one instantiation shows up here)."""
    from lpvmpc import workloads
    for w in (workloads.controller_batch(128, N=20, seed=5), workloads.planner_batch(64, N=30, seed=6),
              workloads.planner_batch(48, N=40, seed=7), workloads.controller_batch(64, N=10, seed=8),
              workloads.controller_batch(64, N=8, seed=9), workloads.planner_batch(64, N=20, seed=10)):
        outs = []
        for variant in (0, 1, 2, 3):
            eng = workloads.make_solver(w)
            eng.set_option("kernel_variant", variant)
            outs.append(eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], w["max_ey"], w["cf_new"], w["lap"]))
            eng.close()
        a = outs[1]
        for b in (outs[0], outs[2], outs[3]):
            assert np.array_equal(a["status"], b["status"]), np.nonzero(a["status"] != b["status"])
            assert np.array_equal(a["iters"], b["iters"]), (np.nonzero(a["iters"] != b["iters"]), a["iters"][a["iters"] != b["iters"]], b["iters"][a["iters"] != b["iters"]])
            assert np.array_equal(a["polish"], b["polish"])
            ok = np.isin(a["status"], (1, 2, -2))
            # polished instances sit on the same optimum; un-polished planner iterates after thousands of iterations carry
            # the (different) round-off of the two elimination orders
            assert np.max(np.abs(a["xPred"][ok] - b["xPred"][ok])) < 5e-6         # observed: 1.5e-6 (N = 40, MFMA against LDS-tile kernel)
            assert np.max(np.abs(a["uPred"][ok] - b["uPred"][ok])) < 5e-6
            pol = ok & (a["polish"] == 1)
            if pol.any():
                assert np.max(np.abs(a["uPred"][pol] - b["uPred"][pol])) < 1e-8


def test_four_wavefront_controller_kernel_matches_the_default(lpvmpc):
    """kernel_variant 9 (the latency form of the controller kernel at N = 20: the two chains relayed over four wavefronts) runs the
    arithmetic of the default kernel step for step; only the block-wide sums associate differently (lpvmpc.h).  Statuses, polish
    flags and iteration counts have to be equal, solutions to 5e-6 -- and on batches whose cost normalisation is decided by the
    linear term's maximum (these) every word is equal, which is asserted for the iterates' outputs.  Handles with straggler
    deferral, steeringDelay > 0 and the other horizons fall back to their default kernels under the option."""
    from lpvmpc import workloads
    for B, seed in ((1, 21), (7, 22), (200, 23), (600, 24)):
        w = workloads.controller_batch(B, N=20, seed=seed)
        outs = []
        for variant in (0, 9):
            eng = workloads.make_solver(w)
            eng.set_option("kernel_variant", variant)
            outs.append(eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]))
            eng.close()
        a, b = outs
        assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["polish"], b["polish"])
        ok = np.isin(a["status"], (1, 2, -2))
        assert np.max(np.abs(a["xPred"][ok] - b["xPred"][ok])) < 5e-6 and np.max(np.abs(a["uPred"][ok] - b["uPred"][ok])) < 5e-6
        assert np.array_equal(a["xPred"], b["xPred"], equal_nan=True) and np.array_equal(a["uPred"], b["uPred"], equal_nan=True)
    # the fall-backs: a deferring handle, a delayed-steering handle and another horizon give their default kernels' words
    w = workloads.controller_batch(300, N=20, seed=25)
    outs = []
    for variant in (0, 9):
        eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant); eng.reserve(300)
        eng.set_option("defer_pool", 600); eng.set_option("defer_after", 100)
        outs.append(eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]))
        eng.close()
    for k in ("xPred", "uPred", "status", "iters", "polish"):
        assert np.array(outs[0][k]).tobytes() == np.array(outs[1][k]).tobytes(), k
    w = workloads.controller_batch(64, N=10, seed=26)
    outs = []
    for variant in (0, 9):
        eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant)
        outs.append(eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]))
        eng.close()
    for k in ("xPred", "uPred", "status", "iters", "polish"):
        assert np.array(outs[0][k]).tobytes() == np.array(outs[1][k]).tobytes(), k
    # the planner's N = 20 kernel has the same latency form (its cost normalisation follows the SUM for this tuning, so the block sums'
    # association shows: decisions equal, solutions to round-off)
    w = workloads.planner_batch(160, N=20, seed=28)
    outs = []
    for variant in (0, 9):
        eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant)
        outs.append(eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"]))
        eng.close()
    a, b = outs
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["polish"], b["polish"])
    ok = np.isin(a["status"], (1, 2, -2))
    assert np.max(np.abs(a["xPred"][ok] - b["xPred"][ok])) < 5e-6 and np.max(np.abs(a["uPred"][ok] - b["uPred"][ok])) < 5e-6
    # the opt-in warm start (row f3) on the latency form: second tick from the first one's shifted (x, y), as the default form takes it
    w = workloads.controller_batch(96, N=20, seed=27)
    outs = []
    for variant in (0, 9):
        eng = workloads.make_solver(w); eng.set_option("kernel_variant", variant); eng.set_option("warm_start", 2)
        o0 = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
        w2 = dict(w)
        w2["x0"] = np.nan_to_num(np.array(o0["xPred"])[:, 1, :]).copy(); w2["u_old"] = np.nan_to_num(np.array(o0["uPred"])[:, 0, :]).copy()
        outs.append(eng.solve(w2["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w2["u_old"], None, w["cf_new"], w["lap"]))
        eng.close()
    a, b = outs
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["polish"], b["polish"])
    ok = np.isin(a["status"], (1, 2, -2))
    assert np.max(np.abs(a["uPred"][ok] - b["uPred"][ok])) < 5e-6


def test_planner_n30_global_scalings_kernel_is_bit_identical(lpvmpc):
    """The planner N = 30 kernel of kernel_variant 7 (round 3's default) keeps its three equilibration vectors in global memory (three instances per CU);
    kernel_variant 5 is the same code with them in LDS (two per CU).  Same arithmetic: every output word has to be equal, for a
    batch that fills the chip unevenly and a ragged one.  (With straggler deferral the handle uses the LDS form -- a parked
    image is the LDS image -- tests/test_gpu_deferral.py compares that against this kernel.)"""
    from lpvmpc import workloads
    for B in (1700, 601):                     # (batches of up to 512 instances take the LDS form anyway)
        w = workloads.planner_batch(B, N=30, seed=14)
        outs = {}
        for variant in (7, 5):
            eng = workloads.make_solver(w)
            eng.set_option("kernel_variant", variant)
            outs[variant] = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
            outs[variant] = {k: np.array(v) for k, v in outs[variant].items() if isinstance(v, np.ndarray)}
            eng.close()
        for k in ("status", "iters", "polish", "xPred", "uPred", "resid"):
            assert np.array_equal(outs[7][k], outs[5][k], equal_nan=True), k
        assert (outs[7]["iters"] > 100).any()


@pytest.mark.gpu
def test_four_wavefront_planner_kernels_are_bit_identical_to_the_two_wavefront_ones(lpvmpc):
    """The default planner kernels at N = 40 / 30 give an instance FOUR wavefronts: the two elimination chains are relayed over two
    wavefronts each and the element phases are cut so that half of them runs beside the sweeps (Solver::kFour).  The arithmetic
    is, step for step, that of the two-wavefront MFMA kernels (kernel_variant 6 / 4): every output word has to be equal -- for a batch
    that fills the chip unevenly (several residencies, stragglers to max_iter, polish) and a ragged one -- and straggler deferral
    (park / restore of the four-wavefront image, bounded passes) has to reproduce the plain call bit for bit."""
    from lpvmpc import workloads
    # (kernel_variant 8 = four wavefronts whatever the batch: the default takes them for launches that leave compute units free and
    # the two-wavefront form for plain launches of 512 instances or more)
    for N, four_wave, two_wave, sizes in ((40, 8, 6, (1100, 37)), (30, 8, 4, (1300, 5))):
        for B in sizes:
            w = workloads.planner_batch(B, N=N, seed=21 + B)
            outs = {}
            for variant in (four_wave, two_wave, 0):
                eng = workloads.make_solver(w)
                eng.set_option("kernel_variant", variant)
                o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
                outs[variant] = {k: np.array(v) for k, v in o.items() if isinstance(v, np.ndarray)}
                eng.close()
            for k in ("status", "iters", "polish", "xPred", "uPred"):
                assert np.array_equal(outs[four_wave][k], outs[two_wave][k], equal_nan=True), (N, B, k)
                assert np.array_equal(outs[0][k], outs[two_wave][k], equal_nan=True), (N, B, k, "default")      # (whichever form the default took for this batch)
            # (residual norms are maxima; the objective value is added up in one order whatever the number of wavefronts: Solver::objective)
            assert np.array_equal(outs[four_wave]["resid"], outs[two_wave]["resid"], equal_nan=True)
            assert np.array_equal(outs[0]["resid"], outs[two_wave]["resid"], equal_nan=True)
            if B > 100:
                assert (outs[four_wave]["iters"] >= 4000).any() and (outs[four_wave]["polish"] == 1).any()
                assert (outs[four_wave]["status"] == -3).any()        # (instances that end on the primal certificate: its u'dy+ + l'dy- sum is taken in one order)
        # Round 6 (advisor): every SUM that reaches a decision is added up in one order in both forms (Solver::kFixedSums), not only the
        # objective.  The default planner tuning hides the cost normalisation's sum -- |q|_inf dominates the mean Hessian column norm, so
        # c = 1 / |q|_inf whatever the sum -- hence a tuning with the linear cost a thousand times smaller (c follows the sum), on a batch
        # with polished, capped and primal-infeasible instances.
        w = dict(workloads.planner_batch(700, N=N, seed=77)); w["L_cf"] = 1e-3 * np.asarray(w["L_cf"])
        outs = {}
        for variant in (four_wave, two_wave):
            eng = workloads.make_solver(w)
            eng.set_option("kernel_variant", variant)
            o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
            outs[variant] = {k: np.array(v) for k, v in o.items() if isinstance(v, np.ndarray)}
            eng.close()
        for k in ("status", "iters", "polish", "xPred", "uPred", "resid"):
            assert np.array_equal(outs[four_wave][k], outs[two_wave][k], equal_nan=True), (N, "small linear cost", k)
        assert (outs[four_wave]["status"] == 1).any() and (outs[four_wave]["status"] == -3).any(), np.unique(outs[four_wave]["status"], return_counts=True)
        # opt-in warm start (shifted) over two ticks, and non-finite inputs (no iteration, NaN out, neighbours untouched): the same on both kernels
        w = workloads.planner_batch(200, N=N, seed=41)
        ticks = {}
        for variant in (four_wave, two_wave):
            eng = workloads.make_solver(w)
            eng.set_option("kernel_variant", variant); eng.set_option("warm_start", 2)
            a = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
            a = {k: np.array(v) for k, v in a.items() if isinstance(v, np.ndarray)}
            x1 = np.where(np.isfinite(a["xPred"][:, 1, :]), a["xPred"][:, 1, :], w["x0"])
            b = eng.solve(x1, w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
            b = {k: np.array(v) for k, v in b.items() if isinstance(v, np.ndarray)}
            bad = w["x0"].copy(); bad[3, 1] = np.nan; bad[77, 0] = np.inf
            eng.set_option("warm_start", 0)
            c = eng.solve(bad, w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
            ticks[variant] = (a, b, {k: np.array(v) for k, v in c.items() if isinstance(v, np.ndarray)})
            eng.close()
        for t in range(3):
            for k in ("status", "iters", "polish", "xPred", "uPred"):
                assert np.array_equal(ticks[four_wave][t][k], ticks[two_wave][t][k], equal_nan=True), (N, "tick", t, k)
        a, b, c = ticks[four_wave]
        assert not np.array_equal(a["iters"], b["iters"])            # (the second tick did run from the first one's state)
        for i in (3, 77):
            assert c["status"][i] == -10 and c["iters"][i] == 0 and np.all(np.isnan(c["xPred"][i])) and np.all(np.isnan(c["uPred"][i]))
        rest = np.ones(200, bool); rest[[3, 77]] = False
        eng = workloads.make_solver(w)
        clean = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
        eng.close()
        assert np.array_equal(c["iters"][rest], np.array(clean["iters"])[rest]) and np.array_equal(c["uPred"][rest], np.array(clean["uPred"])[rest], equal_nan=True)
        # deferral on the four-wavefront kernel: parked after 100 iterations, bounded resume passes of 200, joined
        w = workloads.planner_batch(300, N=N, seed=33)
        eng = workloads.make_solver(w, max_iter=1500)
        plain = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
        plain = {k: np.array(v) for k, v in plain.items() if isinstance(v, np.ndarray)}
        eng.set_option("defer_tail", 0); eng.set_option("defer_after", 100); eng.set_option("defer_budget", 200)
        o = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
        for k in ("status", "iters", "polish", "xPred", "uPred"):
            assert np.array_equal(np.array(o[k]), plain[k], equal_nan=True), (N, "deferral", k)
        assert (plain["iters"] > 300).any()
        eng.close()


def _agree(out, ref, b, nx, tol_x):
    assert int(out["status"][b]) == int(ref["status"][b]), (b, out["status"][b], ref["status"][b], out["iters"][b], ref["iters"][b])
    assert int(out["iters"][b]) == int(ref["iters"][b]), (b, out["iters"][b], ref["iters"][b])
    if np.all(np.isfinite(ref["xPred"][b])):
        assert np.max(np.abs(out["xPred"][b] - ref["xPred"][b])) <= tol_x * max(1.0, np.max(np.abs(ref["xPred"][b]))), b
        assert np.max(np.abs(out["uPred"][b] - ref["uPred"][b])) <= tol_x, b
    else:
        assert np.all(np.isnan(out["xPred"][b])) and np.all(np.isnan(out["uPred"][b]))


def test_fused_batch_cfg3_planner_against_oracle(lpvmpc):
    """BASELINE cfg 3 shape (planner N=30, L-shape, seed 1) at B = 4096: statuses are valid OSQP codes for every
    instance; the first 96 are re-solved by the C oracle tick (same iterations, statuses, solutions)."""
    from lpvmpc import workloads
    w = workloads.planner_batch(4096, N=30, seed=1)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    eng.close()
    assert set(np.unique(out["status"])) <= {1, 2, 3, -2, -3}
    assert np.mean(out["status"] == 1) > 0.7
    ok = out["status"] > 0
    # size-independent properties on the full batch: box feasibility of every returned trajectory within OSQP's eps
    sol = np.isin(out["status"], (1, 2, -2))
    xP, uP = out["xPred"][sol], out["uPred"][sol]
    assert np.all(uP[:, :, 0] <= 0.249 + 5e-3) and np.all(uP[:, :, 0] >= -0.249 - 5e-3)
    assert np.all(uP[:, :, 1] <= 2.0 + 2e-2) and np.all(uP[:, :, 1] >= -0.7 - 2e-2)
    assert np.max(np.abs(xP[:, 0, :] - w["x0"][sol])) <= 5e-3            # x_0 = x0 equality
    n = 96
    sub = {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 4096 else v) for k, v in w.items()}
    ref = O.plan_tick_batch(sub, nthreads=8)
    for b in range(n):
        _agree(out, ref, b, 5, 1e-5)


def test_mixed_cfg4_controller_and_planner_n20(lpvmpc):
    """BASELINE cfg 4 shape: half controller-style, half planner-style instances, both N = 20 (the planner at
    N = 20 runs the run-time-horizon kernel)."""
    from lpvmpc import workloads
    wc = workloads.controller_batch(2048, N=20, seed=2)
    wp = workloads.planner_batch(2048, N=20, seed=2)
    ec, ep = workloads.make_solver(wc), workloads.make_solver(wp)
    oc = ec.solve(wc["x0"], wc["u_prev"], wc["vel_ref"], wc["curv_s"], wc["u_old"], None, wc["cf_new"], wc["lap"])
    op = ep.solve(wp["x0"], wp["u_prev"], None, wp["curv_s"], wp["u_old"], wp["max_ey"])
    ec.close(); ep.close()
    assert np.all(oc["status"] == 1)
    n = 48
    subc = {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 2048 else v) for k, v in wc.items()}
    subp = {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 2048 else v) for k, v in wp.items()}
    rc, rp = O.ctrl_tick_batch(subc, nthreads=8), O.plan_tick_batch(subp, nthreads=8)
    for b in range(n):
        _agree(oc, rc, b, 6, 1e-6)
        _agree(op, rp, b, 5, 1e-5)


@pytest.mark.parametrize("N", [8, 12])
def test_runtime_horizon_kernel_other_N(lpvmpc, N):
    """Horizons without a specialised instantiation (the reference launch file uses N = 8 for the controller,
    MAIN_LAUNCH.launch:117) run the run-time-horizon kernel; checked against the C oracle tick."""
    from lpvmpc import workloads
    w = workloads.controller_batch(96, N=N, seed=11)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    eng.close()
    ref = O.ctrl_tick_batch(w, nthreads=8)
    for b in range(96):
        _agree(out, ref, b, 6, 1e-6)


def test_full_occupancy_runs_are_reproducible(lpvmpc):
    """Two runs of the same 1024-instance batch (every CU holds four instances, two wavefronts each) give bit-identical
    outputs and identical iteration counts.  Regression test for a race in the two-wavefront elimination: wave 0 stored
    x_mid over the middle right-hand side that a delayed wave 1 had not read yet, which only showed under full load as
    iteration counts that changed from run to run."""
    from lpvmpc import workloads
    w = workloads.controller_batch(1024, N=20, seed=0)
    eng = workloads.make_solver(w)
    runs = [eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"]) for _ in range(3)]
    eng.close()
    for r in runs[1:]:
        assert np.array_equal(r["iters"], runs[0]["iters"]) and np.array_equal(r["status"], runs[0]["status"])
        assert np.array_equal(r["uPred"], runs[0]["uPred"]) and np.array_equal(r["xPred"], runs[0]["xPred"])
    ref = O.ctrl_tick_batch(w, nthreads=8)
    assert np.array_equal(runs[0]["iters"], ref["iters"]) and np.array_equal(runs[0]["status"], ref["status"])


def test_planner_n40_batch_against_oracle(lpvmpc):
    """512 planner instances at the launch file's horizon N = 40 (two-wavefront kernel with the largest register
    footprint) against the C oracle tick: statuses, iteration counts, polish outcome by its effect on the solution.
    Regression test for a miscompiled register copy that made the polish step accept wrong candidates."""
    from lpvmpc import workloads
    w = workloads.planner_batch(512, N=40, seed=1)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    eng.close()
    ref = O.plan_tick_batch(w, nthreads=8)
    # the oracle gives up (status -10) on exactly two instances whose roll-out diverged (|A| ~ 1e65); every other instance must
    # agree exactly: no allowance for "threshold sensitive" instances (tests/diagnostics/n40_sensitive.py lists none)
    gave_up = np.nonzero(ref["status"] == -10)[0].tolist()
    assert gave_up == [85, 291], gave_up
    sane = ref["status"] != -10
    assert np.array_equal(out["status"][sane], ref["status"][sane])
    assert np.array_equal(out["iters"][sane], ref["iters"][sane])
    fin = np.isfinite(ref["uPred"]).all(axis=(1, 2)) & sane
    assert np.array_equal(fin, np.isfinite(out["uPred"]).all(axis=(1, 2)) & sane)
    d = np.abs(out["uPred"][fin] - ref["uPred"][fin]).max(axis=(1, 2))
    assert np.max(d) <= 1e-6, (int(np.argmax(d)), float(np.max(d)))


def test_second_device_gets_the_large_lds_opt_in(lpvmpc):
    """The > 64 KB LDS opt-in of a kernel is a per-device function attribute: a handle on device 1 created after one on
    device 0 has to get its own (N = 40 planner: 75 KB).  Needs two visible GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from lpvmpc import workloads
    w = workloads.planner_batch(64, N=40, seed=1)
    outs = []
    for dev in (0, 1):
        eng = workloads.make_solver(w, device=dev)
        outs.append(eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"]))
        eng.close()
    for k in ("status", "iters", "uPred", "xPred"):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=outs[0][k].dtype.kind == "f"), k


def test_maximum_horizon(lpvmpc):
    """LPVMPC_MAX_N = 52 for both problem kinds (largest LDS footprint, run-time-horizon kernel) against the C oracle;
    one more stage is refused at creation."""
    from lpvmpc import workloads
    w = workloads.controller_batch(24, N=52, seed=12)
    eng = workloads.make_solver(w)
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    eng.close()
    ref = O.ctrl_tick_batch(w, nthreads=8)
    for b in range(24):
        _agree(out, ref, b, 6, 1e-6)
    wp = workloads.planner_batch(24, N=52, seed=13)
    eng = workloads.make_solver(wp)
    outp = eng.solve(wp["x0"], wp["u_prev"], None, wp["curv_s"], wp["u_old"], wp["max_ey"])
    eng.close()
    refp = O.plan_tick_batch(wp, nthreads=8)
    assert np.array_equal(outp["status"], refp["status"])
    for b in range(24):
        _agree(outp, refp, b, 5, 1e-6)
    with pytest.raises(lpvmpc.LpvMpcError) as e:
        lpvmpc.BatchedSolver("controller", 53, w["dt"], w["Q"], w["R"], w["dR"], track=w["track"])
    assert e.value.code == -1                                # LPVMPC_E_ARG, not "no device"


def test_empty_and_single_instance_batches(lpvmpc):
    """B = 0 is a no-op with empty outputs; B = 1 equals the first instance of a larger batch bit for bit."""
    from lpvmpc import workloads
    w = workloads.controller_batch(5, N=20, seed=2)
    eng = workloads.make_solver(w)
    e = eng.solve(w["x0"][:0], w["u_prev"][:0], w["vel_ref"][:0], w["curv_s"][:0], w["u_old"][:0], None, w["cf_new"], w["lap"])
    assert e["xPred"].shape == (0, 21, 6) and e["uPred"].shape == (0, 20, 2) and e["status"].shape == (0,)
    S, A, Bm = eng.lpv(w["x0"][:0], w["u_prev"][:0], w["vel_ref"][:0], w["curv_s"][:0])
    assert S.shape == (0, 20, 6) and A.shape == (0, 20, 6, 6)
    assert eng.global_position(np.zeros((0, 2))).shape == (0, 3)
    full = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    one = eng.solve(w["x0"][:1], w["u_prev"][:1], w["vel_ref"][:1], w["curv_s"][:1], w["u_old"][:1], None, w["cf_new"], w["lap"])
    eng.close()
    assert np.array_equal(one["xPred"][0], full["xPred"][0]) and np.array_equal(one["uPred"][0], full["uPred"][0])
    assert one["iters"][0] == full["iters"][0]


def test_small_batch_staging_path_equals_direct_copies(lpvmpc):
    """Host-array calls whose arrays fit 512 KB go through one pinned staging buffer per direction, larger ones use a
    copy per array: the first 48 instances of a 640-instance call (direct copies) equal a 48-instance call (staged) bit
    for bit, for the fused tick, LPVPrediction and solve-with-given-A/B entry points of both problem kinds."""
    from lpvmpc import workloads
    for w in (workloads.controller_batch(640, N=20, seed=31), workloads.planner_batch(640, N=30, seed=32)):
        ctrl = w["kind"] == "controller"
        cut = lambda n: {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 640 and k != "track" else v) for k, v in w.items()}
        eng = workloads.make_solver(w)
        res = []
        for n in (640, 48):
            c = cut(n)
            full = eng.solve(c["x0"], c["u_prev"], c["vel_ref"], c["curv_s"], c["u_old"], c["max_ey"], c["cf_new"], c["lap"])
            S, A, Bm = eng.lpv(c["x0"], c["u_prev"], c["vel_ref"], c["curv_s"], cf_new=c["cf_new"], lap=c["lap"])
            ab = eng.solve_AB(c["x0"], A, Bm, c["vel_ref"], c["u_old"], None if ctrl else c["max_ey"])
            res.append((full, S, A, Bm, ab))
        eng.close()
        (f0, S0, A0, B0, ab0), (f1, S1, A1, B1, ab1) = res
        for k in ("xPred", "uPred", "status", "iters", "polish", "resid"):
            assert np.array_equal(f0[k][:48], f1[k], equal_nan=True), k
            assert np.array_equal(ab0[k][:48], ab1[k], equal_nan=True), k
        assert np.array_equal(S0[:48], S1) and np.array_equal(A0[:48], A1) and np.array_equal(B0[:48], B1)
        # the two-call route (LPVPrediction, then solve with its A / B) is the fused tick
        assert np.array_equal(ab1["iters"], f1["iters"]) and np.array_equal(ab1["uPred"], f1["uPred"], equal_nan=True)


def test_bad_arguments_fail_cleanly(lpvmpc):
    from lpvmpc import workloads
    w = workloads.controller_batch(4, N=20, seed=0)
    with pytest.raises(lpvmpc.LpvMpcError):
        lpvmpc.BatchedSolver("controller", 4, w["dt"], w["Q"], w["R"], w["dR"], track=w["track"])          # N too small
    with pytest.raises(lpvmpc.LpvMpcError):
        lpvmpc.BatchedSolver("controller", 20, w["dt"], w["Q"], w["R"], w["dR"], track=w["track"], rho=-1.0)
    eng = lpvmpc.BatchedSolver("controller", 20, w["dt"], w["Q"], w["R"], w["dR"])                            # no track table
    with pytest.raises(lpvmpc.LpvMpcError):
        eng.solve(w["x0"], w["u_prev"], w["vel_ref"], None, w["u_old"], None, 60.0, 0)                        # lap 0 needs the map
    with pytest.raises(ValueError):
        eng.solve(w["x0"], w["u_prev"][:, :5], w["vel_ref"], w["curv_s"], w["u_old"])                        # wrong shape
    out = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, 60.0, 1)              # lap >= 1 works without a map
    assert np.all(out["status"] == 1)
    eng.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_non_finite_inputs_yield_nan_and_unsolved(lpvmpc, variant):
    """A NaN in one instance's data (a failed measurement, or the NaN solution of an infeasible tick fed back by the
    caller): that instance returns NaN with status UNSOLVED and zero iterations, its neighbours are unaffected.
    (The oracle -- like OSQP, whose max-norms skip NaN -- returns NaN as well, but calls it solved.)"""
    from lpvmpc import workloads
    w = workloads.controller_batch(6, N=20, seed=4)
    eng = workloads.make_solver(w)
    eng.set_option("kernel_variant", variant)
    clean = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])
    x0 = w["x0"].copy(); vel = w["vel_ref"].copy(); uold = w["u_old"].copy()
    x0[1, 3] = np.nan; vel[3, 7] = np.inf; uold[4, 1] = np.nan
    out = eng.solve(x0, w["u_prev"], vel, w["curv_s"], uold, None, w["cf_new"], w["lap"])
    eng.close()
    for b in (1, 3, 4):
        assert out["status"][b] == -10 and out["iters"][b] == 0
        assert np.all(np.isnan(out["xPred"][b])) and np.all(np.isnan(out["uPred"][b]))
    for b in (0, 2, 5):
        assert out["status"][b] == clean["status"][b] and out["iters"][b] == clean["iters"][b]
        assert np.array_equal(out["xPred"][b], clean["xPred"][b]) and np.array_equal(out["uPred"][b], clean["uPred"][b])
    ref = O.ctrl_tick_batch(dict(w, x0=x0))
    assert np.all(np.isnan(ref["uPred"][1]))


def test_opt_in_warm_start_matches_oracle(lpvmpc):
    """Row f3 of SURVEY 8f: warm start from the previous tick's (x, y), shifted by one stage.  Off by default
    (the reference cold-starts every call); when enabled the GPU and the oracle's warm-started OSQP restatement
    take the same iterations and agree on the solution, and fewer iterations are needed than from cold."""
    from lpvmpc import workloads
    B = 128
    w = workloads.controller_batch(B, N=20, seed=21)
    eng = workloads.make_solver(w)
    eng.set_option("warm_start", 2)
    o0 = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], None, w["cf_new"], w["lap"])     # first call: cold
    r0 = O.ctrl_tick_batch(w, nthreads=8)
    assert np.array_equal(o0["iters"], r0["iters"])
    w2 = dict(w)
    w2["x0"] = r0["xPred"][:, 1, :].copy(); w2["u_old"] = r0["uPred"][:, 0, :].copy()
    w2["u_prev"] = np.concatenate([r0["uPred"][:, 1:, :], r0["uPred"][:, -1:, :]], axis=1)
    o1 = eng.solve(w2["x0"], w2["u_prev"], w2["vel_ref"], w2["curv_s"], w2["u_old"], None, w2["cf_new"], w2["lap"])   # warm
    eng.close()
    r1w = O.ctrl_tick_batch(w2, nthreads=8, warm=r0, shift=True)
    r1c = O.ctrl_tick_batch(w2, nthreads=8)
    assert np.array_equal(o1["status"], r1w["status"])
    same = o1["iters"] == r1w["iters"]
    assert same.mean() >= 0.97, (same.mean(), o1["iters"][~same], r1w["iters"][~same])
    pol = same & (o1["polish"] == 1)
    assert np.max(np.abs(o1["xPred"][pol] - r1w["xPred"][pol])) <= 1e-6
    assert np.max(np.abs(o1["uPred"][same] - r1w["uPred"][same])) <= 2e-4
    assert o1["iters"].mean() < 0.85 * r1c["iters"].mean(), (o1["iters"].mean(), r1c["iters"].mean())
    # and the default is still a cold start
    eng = workloads.make_solver(w)
    a = eng.solve(w2["x0"], w2["u_prev"], w2["vel_ref"], w2["curv_s"], w2["u_old"], None, w2["cf_new"], w2["lap"])
    b = eng.solve(w2["x0"], w2["u_prev"], w2["vel_ref"], w2["curv_s"], w2["u_old"], None, w2["cf_new"], w2["lap"])
    eng.close()
    assert np.array_equal(a["iters"], r1c["iters"]) and np.array_equal(b["iters"], r1c["iters"])


def test_full_size_cfg4_properties(lpvmpc):
    """BASELINE configs[3] at full size on one GPU (32768 controller + 32768 planner-style instances, N = 20):
    size-independent properties -- every controller instance solved; every returned trajectory satisfies the
    initial-state equality, the dynamics x_{k+1} = A_k x_k + B_k u_k and the input boxes to OSQP's tolerance; a
    second solve warm-started at its own solution stops at the first termination check with the same answer."""
    from lpvmpc import workloads
    B = 32768
    wc = workloads.controller_batch(B, N=20, seed=2)
    ec = workloads.make_solver(wc)
    S, A, Bm = ec.lpv(wc["x0"], wc["u_prev"], wc["vel_ref"], wc["curv_s"], cf_new=60.0, lap=1)
    oc = ec.solve(wc["x0"], wc["u_prev"], wc["vel_ref"], wc["curv_s"], wc["u_old"], None, 60.0, 1)
    # a handful of extreme random states (low speed in a curve) run into max_iter / "solved inaccurate", as in OSQP
    assert set(np.unique(oc["status"])) <= {1, 2, -2} and np.mean(oc["status"] == 1) > 0.999
    s1 = oc["status"] == 1
    x, u = oc["xPred"][s1], oc["uPred"][s1]
    # every constraint violation of the returned point is bounded by the primal residual the solver reports
    # (||Ax - z||_inf with z inside the bounds), which OSQP's stopping rule keeps below eps_abs + eps_rel max(|Ax|, |z|)
    pri = oc["resid"][s1, 0]
    scale = np.maximum(np.max(np.abs(x), axis=(1, 2)), 1.0)
    assert np.all(pri <= 1e-3 + 1e-3 * 1.5 * scale)
    tol = 1.01 * pri + 1e-9
    dyn = x[:, 1:, :] - np.einsum("bkij,bkj->bki", A[s1], x[:, :-1, :]) - np.einsum("bkij,bkj->bki", Bm[s1], u)
    assert np.all(np.max(np.abs(x[:, 0, :] - wc["x0"][s1]), axis=1) <= tol)
    assert np.all(np.max(np.abs(dyn), axis=(1, 2)) <= tol)
    assert np.all(np.max(np.abs(u[:, :, 0]), axis=1) <= 0.249 + tol)
    assert np.all(np.max(u[:, :, 1], axis=1) <= 4.0 + tol) and np.all(np.min(u[:, :, 1], axis=1) >= -1.0 - tol)
    assert np.all(np.min(x[:, :-1, 0], axis=1) >= 0.01 - tol)
    # idempotence under a warm start at the solution
    ec.set_option("warm_start", 1)
    ec.solve(wc["x0"], wc["u_prev"], wc["vel_ref"], wc["curv_s"], wc["u_old"], None, 60.0, 1)           # stores the state
    o2 = ec.solve(wc["x0"], wc["u_prev"], wc["vel_ref"], wc["curv_s"], wc["u_old"], None, 60.0, 1)      # warm at its own solution
    ec.close()
    assert np.mean(o2["status"] == 1) > 0.999 and np.mean(o2["iters"][s1] == 25) > 0.99
    pol = (oc["polish"] == 1) & (o2["polish"] == 1)
    # an accepted polish with a mis-guessed active set is still an accepted polish (OSQP semantics), so "same optimum"
    # holds for almost all, not all, instances
    d = np.max(np.abs(o2["uPred"][pol] - oc["uPred"][pol]), axis=(1, 2))
    assert np.mean(d <= 1e-6) > 0.99 and np.max(d) <= 5e-2
    wp = workloads.planner_batch(B, N=20, seed=2)
    ep = workloads.make_solver(wp)
    op = ep.solve(wp["x0"], wp["u_prev"], None, wp["curv_s"], wp["u_old"], wp["max_ey"])
    ep.close()
    assert set(np.unique(op["status"])) <= {1, 2, 3, -2, -3}
    sol = np.isin(op["status"], (1, 2, -2))
    assert sol.mean() > 0.7
    assert np.all(np.isnan(op["xPred"][~sol])) and np.all(np.isfinite(op["xPred"][sol]))
    assert np.max(np.abs(op["xPred"][sol][:, 0, :] - wp["x0"][sol])) <= 5e-3


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_controller_steering_delay_matches_reference_fixture(lpvmpc, variant):
    """the pinned-steering equality rows of _buildMatEqConst (CTRL:518-527; OldSteering of length 1 + delay,
    CTRL:71) -- fixture generated with the reference's own class at steeringDelay = 1, 2, 3; every kernel variant."""
    tab = lpvmpc.Map("oval", 0.2).PointAndTangent
    for i, c in enumerate(cases("ctrl_n20_delay")):
        N = int(c["N"]); d = int(c["steer_hist"].size)
        eng = lpvmpc.BatchedSolver("controller", N, float(c["dt"]), c["Q"], c["R"], c["dR"], track=tab, steering_delay=d)
        eng.set_option("kernel_variant", variant)
        u_old = np.concatenate([c["old_u"], c["steer_hist"]])[None]
        out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], vfull(c)[None], u_old)
        eng.close()
        assert int(out["status"][0]) == int(c["status_orc"]), (i, out["status"][0])
        assert int(out["iters"][0]) == int(c["iter_orc"]), (i, out["iters"][0], c["iter_orc"])
        assert int(out["polish"][0]) == int(c["polish_orc"]), i
        tol = 1e-6 if int(c["polish_orc"]) == 1 else 2e-4
        relclose(out["xPred"][0], c["xPred"], tol)
        assert np.max(np.abs(out["uPred"][0] - c["uPred"])) <= tol, i
        assert np.max(np.abs(out["uPred"][0][:d, 0] - c["steer_hist"])) <= 2e-4          # the pins hold
        z = np.concatenate([out["xPred"][0].reshape(-1), out["uPred"][0].reshape(-1)])
        ok, info = kkt_cert.osqp_termination_ok(c["P"], c["q"], c["Aqp"], np.where(c["l"] < -1e29, -np.inf, c["l"]), c["u"], z, c["y_orc"])
        assert ok, info


def test_dropin_controller_with_steering_delay(lpvmpc):
    """The drop-in class with steeringDelay = 2: OldSteering has three entries (CTRL:71) and solve() pins u_0, u_1."""
    c = cases("ctrl_n20_delay")[1]
    assert c["steer_hist"].size == 2
    mp = lpvmpc.Map("oval", 0.2)
    ctl = lpvmpc.PathFollowingLPV_MPC(c["Q"], c["R"], c["dR"], int(c["N"]), 1, float(c["dt"]), mp, "OSQP", 2, 0)
    assert len(ctl.OldSteering) == 3 and len(ctl.OldPredicted) == 1 + 2 + int(c["N"])
    ctl.OldSteering = [float(c["old_u"][0])] + [float(v) for v in c["steer_hist"]]; ctl.OldAccelera = [float(c["old_u"][1])]
    S, A_L, B_L, C_L = ctl.LPVPrediction(c["x0"], c["u_prev"], c["vel_ref"], c["curv_ref"], float(c["cf_new"]), int(c["lap"]))
    ctl.solve(c["x0"], 0.0, c["u_prev"], False, c["vel_ref"], A_L, B_L, C_L, 10)
    assert ctl.iters == int(c["iter_orc"]) and ctl.status_val == 1
    relclose(ctl.xPred, c["xPred"], 1e-6); relclose(ctl.uPred, c["uPred"], 1e-6)


# ---- what bounds the unpinned OSQP stage ---------------------------------------------------
def _bounds(c):
    return np.where(c["l"] < -1e29, -np.inf, c["l"]), np.where(c["u"] > 1e29, np.inf, c["u"])


@pytest.mark.parametrize("name", ["plan_n30_lshape", "plan_n40_lshape"])
def test_planner_point_passes_osqp_stopping_rule_and_objective_gap(lpvmpc, name):
    """The planner's returned point, checked without the oracle's solver: OSQP's termination test re-evaluated in numpy on the
    device's primal point (with the multipliers of the same iteration) and the objective against the KKT-certified optimum
    of the golden case (relative gap <= 1e-3, one-sided: an eps = 1e-3 iterate may undercut the optimum from slightly outside
    the feasible set).  Until round 2 only the controller had this check."""
    tab = lpvmpc.Map("L_shape", 0.2).PointAndTangent
    checked = 0
    for i, c in enumerate(cases(name)):
        N = int(c["N"])
        eng = lpvmpc.BatchedSolver("planner", N, float(c["dt"]), c["Q"], c["R"], c["dR"], L_cf=c["L_cf"], track=tab)
        out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], None, np.zeros((1, 2)), max_ey=float(c["max_ey"]))
        eng.close()
        if int(out["status"][0]) != 1:
            continue
        l, u = _bounds(c)
        z = np.concatenate([out["xPred"][0].reshape(-1), out["uPred"][0].reshape(-1)])
        ok, info = kkt_cert.osqp_termination_ok(c["P"], c["q"], c["Aqp"], l, u, z, c["y_orc"])
        assert ok, (name, i, info)
        xs = c["x_star"]
        if np.all(np.isfinite(xs)):
            f = lambda v: 0.5 * v @ c["P"] @ v + c["q"] @ v
            assert f(z) - f(xs) <= 1e-3 * max(1.0, abs(f(xs))), (name, i, f(z), f(xs))
            checked += 1
    assert checked >= 3


@pytest.mark.parametrize("interval", [25, 50, 100])
def test_rho_update_interval_sensitivity_on_the_device(lpvmpc, interval):
    """adaptive_rho_interval is the one OSQP default that is machine dependent in the wheel (0 = from the measured set-up
    time).  For 25 / 50 / 100 the device agrees with the oracle run at the same setting (status, iterations) and its result
    stays inside the stated float tolerance: polished controller solves within 1e-3 of the KKT-certified optimum, un-polished
    ones and the planner within OSQP's stopping rule and a 1e-3 relative objective gap (tests/test_oracle_osqp.py holds the
    same test for the oracle alone)."""
    from tests.test_oracle_osqp import check_against_optimum
    oval, lsh = lpvmpc.Map("oval", 0.2).PointAndTangent, lpvmpc.Map("L_shape", 0.2).PointAndTangent
    for name in ("ctrl_n10_cfg1", "ctrl_n20_oval", "ctrl_n20_delay"):
        for i, c in enumerate(cases(name)):
            N = int(c["N"])
            hist = np.asarray(c["steer_hist"], float).reshape(-1) if "steer_hist" in c else np.zeros(0)
            eng = lpvmpc.BatchedSolver("controller", N, float(c["dt"]), c["Q"], c["R"], c["dR"], track=oval,
                                       steering_delay=len(hist), adaptive_rho_interval=interval)
            u_old = np.concatenate([np.asarray(c["old_u"], float).reshape(-1), hist])[None]
            out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], vfull(c)[None], u_old)
            eng.close()
            r = O.solve_qp(c["P"], c["q"], c["Aqp"], c["l"], c["u"], adaptive_rho_interval=interval)
            assert (int(out["status"][0]), int(out["iters"][0]), int(out["polish"][0])) == (r.info.status_val, r.info.iter, r.info.status_polish), (name, i, interval)
            z = np.concatenate([out["xPred"][0].reshape(-1), out["uPred"][0].reshape(-1)])
            check_against_optimum(c, z, r.y, int(out["polish"][0]) == 1, 6, (name, i, interval))
    for name in ("plan_n30_lshape", "plan_n40_lshape"):
        for i, c in enumerate(cases(name)):
            N = int(c["N"])
            eng = lpvmpc.BatchedSolver("planner", N, float(c["dt"]), c["Q"], c["R"], c["dR"], L_cf=c["L_cf"], track=lsh,
                                       adaptive_rho_interval=interval)
            out = eng.solve_AB(c["x0"][None], c["A"][None], c["B"][None], None, np.zeros((1, 2)), max_ey=float(c["max_ey"]))
            eng.close()
            r = O.solve_qp(c["P"], c["q"], c["Aqp"], c["l"], c["u"], adaptive_rho_interval=interval)
            assert (int(out["status"][0]), int(out["iters"][0])) == (r.info.status_val, r.info.iter), (name, i, interval)
            if r.info.status_val == 1 and np.all(np.isfinite(c["x_star"])):
                z = np.concatenate([out["xPred"][0].reshape(-1), out["uPred"][0].reshape(-1)])
                check_against_optimum(c, z, r.y, False, 5, (name, i, interval))


def test_curvature_lookup_failure_reaches_the_caller(lpvmpc):
    """Where the reference's Curvature() raises (UTIL:44-48: no segment contains s -- s < 0, s exactly at the end of the closing
    segment) the device lookup yields NaN: the LPV blocks are NaN, no iteration is run, the instance comes back UNSOLVED with
    NaN outputs, its neighbours in the batch are untouched.  A far-ahead or non-finite abscissa (diverged roll-out) ends the same
    way instead of spinning in the wrap loop."""
    from lpvmpc import workloads
    w = workloads.planner_batch(6, N=30, seed=5)
    L_track = float(w["track"][-1, 3] + w["track"][-1, 4])
    SS = w["curv_s"].copy()
    SS[1, 3] = -0.5                      # no segment: the reference raises
    SS[2, 0] = L_track                   # exactly the track length: not wrapped (s > L is false), not inside the last segment
    SS[3, 7] = 1e12                      # would take 5e10 subtractions
    SS[4, 2] = np.inf
    eng = workloads.make_solver(w)
    S, A, Bm = eng.lpv(w["x0"], w["u_prev"], None, SS)
    out = eng.solve(w["x0"], w["u_prev"], None, SS, w["u_old"], w["max_ey"])
    ref = eng.solve(w["x0"], w["u_prev"], None, w["curv_s"], w["u_old"], w["max_ey"])
    eng.close()
    for b in (1, 2, 3, 4):
        assert not np.all(np.isfinite(A[b])), b
        assert int(out["status"][b]) == -10 and int(out["iters"][b]) == 0, (b, out["status"][b])
        assert np.all(np.isnan(out["uPred"][b])) and np.all(np.isnan(out["xPred"][b]))
    for b in (0, 5):
        assert np.all(np.isfinite(A[b]))
        assert np.array_equal(out["uPred"][b], ref["uPred"][b], equal_nan=True) and out["status"][b] == ref["status"][b]
