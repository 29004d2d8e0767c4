import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import lpvmpc
from lpvmpc import workloads
from oracle import lpv_ref as L
from tests._golden import load
g = load("handoff")
Np, dtp = int(g["N"]), float(g["dt"])
mp = lpvmpc.Map("L_shape", 0.2)
for variant in (0, 3):
    p = lpvmpc.LPV_MPC_Planner(workloads.PLAN_Q, workloads.PLAN_R, workloads.PLAN_dR, workloads.PLAN_L, Np, dtp, mp, "OSQP")
    p._eng.set_option("kernel_variant", variant)
    ho = lpvmpc.PlannerHandoff(p)
    x0 = np.array([1.0, 0.0, 0.0, 0.03, -0.02]); first = 1; errs = []
    for tick in range(20):
        if first == 1:
            xx, uu = L.plan_seed_vectors(Np, x0, 0.2, dtp); p.solve(x0, xx, uu, 0, 0, 0, first, 0.2); first += 1
        else:
            S, A_L, B_L, C_L = p.LPVPrediction(p.xPred[1, :], ho.SS, p.uPred); p.solve(p.xPred[1, :], 0, 0, A_L, B_L, C_L, first, 0.2)
        p.OldSteering.append(p.uPred[0, 0]); p.OldAccelera.append(p.uPred[0, 1])
        refs = ho.update()
        errs.append((int(p.iters == g["plan_iters"][tick]), float(np.max(np.abs(p.xPred - g["plan_xPred"][tick]))), float(np.max(np.abs(refs - g["plan_refs"][tick])))))
    print("variant", variant, " ".join("%d:%.1e/%.1e" % e for e in errs))
