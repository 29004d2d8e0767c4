#!/usr/bin/env python3
"""The table that opens DESIGN.md section 4 -- which solve kernel ships for which (kind, N) -- from the device assembly the build keeps
(autonomous-racing-lpv-mpp-mpc_amd/csrc/build/admm_solve-*.s: registers, scratch, instructions between the hot-path markers) and the
kernel's own LDS formula (Solver::lds_doubles).  Usage: python3 tools/kernel_table.py [file.s]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_kernel_resources as ckr

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "autonomous-racing-lpv-mpp-mpc_amd", "csrc", "build", "admm_solve-hip-amdgcn-amd-amdhsa-gfx950.s")
kTS = 72


def lds_bytes(N, fixn, gs, tail, nw):
    red = 160 if (nw == 4 and not tail) else 80
    img = (N + 1) * ((1 if fixn else 3) * kTS + (16 if gs else 19) * 8 + 8) + 16 + 64 + 8 + red + 64
    if nw == 4 and not tail and 2 * (N + 1) * 8 < 448:
        img += 448        # kFourSmall: the factorisation's hand-over area
    if tail:
        img += (N + 1) * (2 * kTS + 7 * 64) + 96 + nw * 128
    return img * 8


# (kind, N, kernel_variant, note) -> template arguments <NX, NT, NW, MF, GS, TAIL>
SHIPS = [
    ("controller", 20, "0 (default)", (6, 20, 2, 1, 0, 0), "MFMA sweeps + factorisation, two wavefronts (twisted elimination)"),
    ("controller", 20, "9 (latency form: one vehicle per handle, the drop-in class)", (6, 20, 4, 1, 0, 0), "the default's arithmetic, chains relayed over four wavefronts"),
    ("controller", 20, "tail (closing passes of the deferral, `defer_tail` 1)", (6, 20, 8, 0, 0, 1), "whole-CU kernel: dense K^-1 in registers, two-phase iteration, checks off the chain"),
    ("controller", 20, "3", (6, 20, 2, 0, 0, 0), "DPP sweeps, two wavefronts (round 1)"),
    ("controller", 20, "2", (6, 20, 1, 0, 0, 0), "one wavefront, factor in registers"),
    ("controller", 10, "0", (6, 10, 1, 0, 0, 0), "one wavefront, factor in registers"),
    ("controller", 8, "0", (6, 8, 1, 0, 0, 0), "one wavefront, factor in registers (the launch file's horizon)"),
    ("controller", "other <= 52", "0 / 1", (6, 0, 1, 0, 0, 0), "run-time horizon, factor tiles in LDS"),
    ("planner", 20, "0", (5, 20, 2, 1, 0, 0), "MFMA sweeps + factorisation, two wavefronts"),
    ("planner", 20, "9 (latency form)", (5, 20, 4, 1, 0, 0), "the default's arithmetic, chains relayed over four wavefronts"),
    ("planner", 20, "tail", (5, 20, 8, 0, 0, 1), "whole-CU kernel (as the controller's)"),
    ("planner", 20, "3", (5, 20, 2, 0, 0, 0), "DPP sweeps, two wavefronts"),
    ("planner", 30, "0 (B < 512, or deferral) / 8", (5, 30, 4, 1, 0, 0), "MFMA sweeps, chains relayed over four wavefronts"),
    ("planner", 30, "7 (B > 512, no deferral)", (5, 30, 2, 0, 1, 0), "DPP sweeps, equilibration vectors in global memory (three per CU)"),
    ("planner", 30, "5 / 7 / 3", (5, 30, 2, 0, 0, 0), "DPP sweeps, two wavefronts"),
    ("planner", 30, "0 (plain launch, B >= 512) / 4", (5, 30, 2, 1, 0, 0), "MFMA sweeps, two wavefronts (the same arithmetic as the four-wavefront form)"),
    ("planner", 30, "2", (5, 30, 1, 0, 0, 0), "one wavefront"),
    ("planner", 40, "0 / 8", (5, 40, 4, 1, 0, 0), "MFMA sweeps, chains relayed over four wavefronts"),
    ("planner", 40, "6", (5, 40, 2, 1, 0, 0), "MFMA sweeps, two wavefronts (round 3's default; the same arithmetic as the four-wavefront form)"),
    ("planner", 40, "3", (5, 40, 2, 0, 0, 0), "DPP sweeps, two wavefronts"),
    ("planner", "other <= 52", "0 / 1", (5, 0, 1, 0, 0, 0), "run-time horizon, factor tiles in LDS"),
]

ks = {k["name"]: k for k in ckr.kernels(path)}
hot = ckr.hot_path(path)
print("| problem | `kernel_variant` | instantiation `<NX, NT, NW, MF, GS, TAIL>` | what it is | wavefronts | VGPRs | scratch (B) | LDS per instance | instances per CU | instructions of an iteration (all role and round paths, reachable between the markers) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for kind, N, variant, targs, what in SHIPS:
    nx, nt, nw, mf, gs, tail = targs
    mangled = "ILi%dELi%dELi%dELb%dELb%dELb%dE" % (nx, nt, nw, mf, gs, tail)
    name = next((n for n in ks if mangled in n), None)
    k = ks.get(name, {})
    n_for_lds = nt if nt else 20
    lds = lds_bytes(n_for_lds, nt > 0, gs, tail, nw)
    vg = k.get("vgpr_count", -1)
    waves_per_simd = 1 if vg > 256 else 2
    by_regs = (4 * waves_per_simd * (512 // max(vg, 1) if False else (2 if vg <= 256 else 1))) // nw if nw else 0
    by_lds = (160 * 1024) // lds
    per_cu = max(1, min(by_lds, (8 if vg <= 256 else 4) // nw)) if not tail else 1
    h = hot.get(name, {})
    print("| %s N = %s | %s | `<%d, %d, %d, %d, %d, %d>` | %s | %d | %d | %d | %.1f KB%s | %d | %s |" % (
        kind, N, variant, nx, nt, nw, mf, gs, tail, what, nw, vg, k.get("private_segment_fixed_size", 0), lds / 1024.0,
        " (at N = 20)" if not nt else "", per_cu, h.get("instructions", "-")))
