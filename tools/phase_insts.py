#!/usr/bin/env python3
"""Instruction counts of the PHASES OF A WHOLE SOLVE (set-up, Ruiz scaling, factorisation, iteration, check, polish) of the headline
kernel by differencing launches with different settings under a counter pass:
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -o c -- python3 tools/phase_insts.py
then  python3 tools/phase_insts.py --summary <dir>.  Every configuration is ONE launch of 1024 controller instances (N = 20, seed 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = 1024
base = dict(adaptive_rho=0, polish=0, check_termination=0)
CONFIGS = [
    ("iter1", dict(max_iter=1, **base)),
    ("iter1_noscale", dict(max_iter=1, scaling=0, **base)),
    ("iter101", dict(max_iter=101, **base)),
    ("iter201", dict(max_iter=201, **base)),
    ("polish50", dict(max_iter=50, adaptive_rho=0, polish=1, check_termination=50, eps_abs=1e3, eps_rel=1e3)),
    ("nopolish50", dict(max_iter=50, adaptive_rho=0, polish=0, check_termination=50, eps_abs=1e3, eps_rel=1e3)),
    ("refactor100", dict(max_iter=100, adaptive_rho=1, adaptive_rho_interval=25, adaptive_rho_tolerance=1.0000001, polish=0, check_termination=0)),
    ("plain100", dict(max_iter=100, **base)),
    ("checks100", dict(max_iter=100, adaptive_rho=0, polish=0, check_termination=25, eps_abs=1e-12, eps_rel=1e-12)),
    ("default", dict()),
]
if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    import csv, glob
    f = glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True)[0]
    rows = {}
    for r in csv.DictReader(open(f)):
        if "admm_solve_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) < B * 128: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)[-len(CONFIGS):]
    c = {name: rows[i] for (name, _), i in zip(CONFIGS, ids)}
    def d(a, b, div=1.0): return {k: (c[a][k] - (c[b][k] if b else 0.0)) / (B * div) for k in c[a]}
    def line(label, v): print("%-58s" % label + "  ".join("%s %9.0f" % (k.replace("SQ_", "").replace("INSTS_", ""), v[k]) for k in sorted(v)))
    print("per instance (wave-instructions of both wavefronts; cycles summed over the waves):")
    line("one ADMM iteration", d("iter201", "iter101", 100.0))
    line("Ruiz scaling (10 passes)", d("iter1", "iter1_noscale"))
    line("load, first factorisation, 1 iteration, final check, write", d("iter1_noscale", None))
    line("one rho update (residuals + factorisation + weights)", d("refactor100", "plain100", 4.0))
    line("one termination check (residuals + tests)", d("checks100", "plain100", 4.0))
    line("polish (factorisation, 1 + 3 solves, products, residuals)", d("polish50", "nopolish50"))
    line("a default solve, whole", d("default", None))
    sys.exit(0)
import numpy as np
from lpvmpc import workloads
w = workloads.controller_batch(B, N=20, seed=0)
for name, settings in CONFIGS:
    eng = workloads.make_solver(w, **settings)
    o = eng.solve(w["x0"], w["u_prev"], w["vel_ref"], w["curv_s"], w["u_old"], w["max_ey"], w["cf_new"], w["lap"])
    print(name, "iterations mean %.1f" % o["iters"].mean(), "status", np.unique(o["status"]), flush=True)
    eng.close()
