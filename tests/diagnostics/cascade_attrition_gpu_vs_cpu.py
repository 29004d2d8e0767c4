import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from tests._golden import load
from tests.test_gpu_cascade import planner, controller_tt, fleet_start
from oracle import cascade_ref as CR
from lpvmpc import workloads as W
c = load("cascade")
B, K = 32, 90
plant0 = fleet_start(c, 3, B, spread=0.01)
cmd0 = np.tile(c["cmd0"], (B, 1)); uPred0 = np.tile(c["uPred0"], (B, 1, 1))
plan, mp = planner(); plan.handoff_setup(); ctrl = controller_tt(mp)
ctrl.cascade_init(plan, plant0, cmd0, uPred0, lap0=1, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2, q9_swap=True)
ref = CR.CascadeRef(mp.PointAndTangent, W.CTRL_TUNINGS["race"], (W.PLAN_Q, W.PLAN_R, W.PLAN_dR, W.PLAN_L), plant0, cmd0, uPred0, half_width=mp.halfWidth, slack=mp.slack, plan_max_ey=0.2, nthreads=16)
dd = np.full(B, -1); dr = np.full(B, -1)
for k in range(K):
    ctrl.cascade_tick(1); ref.tick()
    o = ctrl.cascade_read(full=False)
    a = np.all(np.isfinite(o["plant"]), axis=1); r = np.all(np.isfinite(ref.plant), axis=1)
    dd[(dd < 0) & ~a] = k; dr[(dr < 0) & ~r] = k
    both = a & r
    if k % 10 == 9: print(k, "alive dev %d ref %d; survivors max|dplant| %.2e; plan iters equal %d/%d" % (a.sum(), r.sum(), np.max(np.abs(o["plant"][both]-ref.plant[both])), np.sum(o["plan_iters"][both]==ref.plan["iters"][both]), both.sum()))
print("dev", {int(b): int(dd[b]) for b in np.nonzero(dd>=0)[0]})
print("ref", {int(b): int(dr[b]) for b in np.nonzero(dr>=0)[0]})
