"""CPU: the C-ABI shared library loads and exports every symbol include/lpvmpc.h declares; no compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lpvmpc.h")


@pytest.fixture(scope="module")
def ffi():
    from lpvmpc import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "autonomous-racing-lpv-mpp-mpc_amd", "csrc")], check=True)
    return _ffi


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lpvmpc_[a-z_A-Z]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(ffi):
    lib = ffi.load()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "liblpvmpc.so does not export %s" % n
    assert set(ffi.EXPORTS) == set(names)


def test_version_and_struct_layout(ffi, tmp_path):
    lib = ffi.load()
    assert lib.lpvmpc_version() == 200
    src = tmp_path / "sz.c"
    src.write_text('#include "lpvmpc.h"\n#include <stdio.h>\n#include <stddef.h>\n'
                   'int main(){printf("%zu %zu %zu %zu\\n", sizeof(lpvmpc_config), offsetof(lpvmpc_config, Q),'
                   ' offsetof(lpvmpc_config, rho), offsetof(lpvmpc_config, track)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, oq, orho, otrack = map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert C.sizeof(ffi.Config) == size
    assert (ffi.Config.Q.offset, ffi.Config.rho.offset, ffi.Config.track.offset) == (oq, orho, otrack)
    src.write_text('#include "lpvmpc.h"\n#include <stdio.h>\n#include <stddef.h>\n'
                   'int main(){printf("%zu %zu %zu %zu\\n", sizeof(lpvmpc_handoff_config), offsetof(lpvmpc_handoff_config, order),'
                   ' offsetof(lpvmpc_handoff_config, b), offsetof(lpvmpc_handoff_config, a)); return 0;}\n')
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, oo, ob, oa = map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    H = ffi.HandoffConfig
    assert (C.sizeof(H), H.order.offset, H.b.offset, H.a.offset) == (size, oo, ob, oa)


def test_default_config_matches_reference_constants(ffi):
    c = ffi.default_config(ffi.KIND_CONTROLLER)
    assert (c.N, c.lf, c.lr, c.m, c.Iz, c.Cf, c.Cr, c.mu) == (20, 0.125, 0.125, 1.98, 0.03, 60.0, 60.0, 0.05)   # MAIN_LAUNCH.launch:5-11
    assert (c.ctrl_vx_min, c.ctrl_delta_max, c.ctrl_a_max, c.ctrl_a_min_abs) == (0.01, 0.249, 4.0, 1.0)         # CTRL:334-348
    assert (c.rho, c.sigma, c.alpha, c.eps_abs, c.eps_rel, c.max_iter, c.check_termination, c.scaling, c.polish) == \
           (0.1, 1e-6, 1.6, 1e-3, 1e-3, 4000, 25, 10, 1)
    p = ffi.default_config(ffi.KIND_PLANNER)
    assert (p.N, p.dt, p.max_vel, p.min_vel) == (30, 0.05, 5.0, 0.9)
    assert list(p.plan_umin) == [-0.249, -0.7] and list(p.plan_umax) == [0.249, 2.0]                            # PLAN:173-174
    assert p.Q[3 * 5 + 3] < 0                                                                                   # quirk Q4


def test_no_device_fails_loudly(ffi):
    """Without a HIP device lpvmpc_create must fail with a message -- there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    import lpvmpc
    with pytest.raises(lpvmpc.LpvMpcError) as e:
        lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, np.eye(6), np.eye(2), np.ones(2))
    assert e.value.code == ffi.E_NODEVICE and "no CPU fallback" in str(e.value)
    m = lpvmpc.Map("oval")
    with pytest.raises(lpvmpc.LpvMpcError):
        lpvmpc.PathFollowingLPV_MPC(np.eye(6), np.eye(2), np.ones(2), 20, 1, 1 / 30.0, m, "OSQP", 0, 0)
    with pytest.raises(lpvmpc.LpvMpcError) as e:            # steeringDelay is supported: the failure is the missing device
        lpvmpc.PathFollowingLPV_MPC(np.eye(6), np.eye(2), np.ones(2), 20, 1, 1 / 30.0, m, "OSQP", 3, 0)
    assert e.value.code == ffi.E_NODEVICE
    with pytest.raises(lpvmpc.LpvMpcError) as e:            # argument checks come before the device probe
        lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, np.eye(6), np.eye(2), np.ones(2), steering_delay=9)
    assert e.value.code == ffi.E_ARG and "steering_delay" in str(e.value)


def test_device_assembly_has_no_copy_in_front_of_an_exec_restore(ffi):
    """Build-time guard, repeated here: the device assembly kept by the Makefile (csrc/build/*.s) must not contain a
    whole-register copy in front of the `s_or_b64 exec` that re-opens the mask at a control-flow join (an LLVM
    machine-sink placement that corrupted a live register of the N = 40 planner kernel)."""
    import glob
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_exec_prologue as chk
    build = os.path.join(ROOT, "autonomous-racing-lpv-mpp-mpc_amd", "csrc", "build")
    files = sorted(glob.glob(os.path.join(build, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not files:
        subprocess.run(["make", "-B", "-C", os.path.dirname(build)], check=True)
        files = sorted(glob.glob(os.path.join(build, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    assert len(files) >= 4
    for f in files:
        assert chk.scan(f) == [], f
    # the scanner does recognise the pattern
    bad = os.path.join(str(build), "_synthetic.s")
    with open(bad, "w") as fh:
        fh.write("_Zk:\n.LBB0_1:\n\tv_mov_b32_e32 v200, v188\n\ts_mov_b64 s[74:75], s[70:71]\n\ts_or_b64 exec, exec, s[12:13]\n\tds_write_b64 v1, v[2:3]\n")
    try:
        assert len(chk.scan(bad)) == 1
        # ... and does not mistake the tail of a guarded region opened inside the block (default for all lanes, narrowed mask,
        # copy for the lanes of the region, restore) for it
        with open(bad, "w") as fh:
            fh.write("_Zk:\n.LBB0_1:\n\tv_mov_b32_e32 v33, 0x58\n\ts_and_saveexec_b64 s[12:13], s[20:21]\n\tds_read_b64 v[30:31], v233\n"
                     "\tv_mov_b32_e32 v33, v234\n\ts_or_b64 exec, exec, s[12:13]\n\tds_write_b64 v1, v[2:3]\n")
        assert chk.scan(bad) == []
        # ... but a copy behind the narrowing whose destination got NO value under the wider mask is still flagged (the lanes outside
        # the region would reach the join with a stale register), and so is a copy behind a mask-WIDENING instruction
        with open(bad, "w") as fh:
            fh.write("_Zk:\n.LBB0_1:\n\ts_and_saveexec_b64 s[12:13], s[20:21]\n\tds_read_b64 v[30:31], v233\n"
                     "\tv_mov_b32_e32 v33, v234\n\ts_or_b64 exec, exec, s[12:13]\n\tds_write_b64 v1, v[2:3]\n")
        assert len(chk.scan(bad)) == 1
        with open(bad, "w") as fh:
            fh.write("_Zk:\n.LBB0_1:\n\tv_mov_b32_e32 v33, 0x58\n\ts_or_saveexec_b64 s[12:13], s[20:21]\n"
                     "\tv_mov_b32_e32 v33, v234\n\ts_or_b64 exec, exec, s[12:13]\n\tds_write_b64 v1, v[2:3]\n")
        assert len(chk.scan(bad)) == 1
        # the else side of a structurised if / else (s_or_saveexec + s_xor exec): its phi copy is legitimate when the then side, in
        # front of the label, wrote the same register for its own lanes -- and flagged when it did not
        with open(bad, "w") as fh:
            fh.write("_Zk:\n\tv_lshlrev_b32_e32 v46, 3, v1\n\tds_write_b64 v46, v[2:3]\n.LBB0_1:\n\ts_or_saveexec_b64 s[0:1], s[2:3]\n"
                     "\tv_lshlrev_b32_e32 v30, 3, v1\n\ts_xor_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v46, v30\n\ts_or_b64 exec, exec, s[0:1]\n\tds_write_b64 v1, v[2:3]\n")
        assert chk.scan(bad) == []
        with open(bad, "w") as fh:
            fh.write("_Zk:\n\tv_lshlrev_b32_e32 v45, 3, v1\n\tds_write_b64 v45, v[2:3]\n.LBB0_1:\n\ts_or_saveexec_b64 s[0:1], s[2:3]\n"
                     "\tv_lshlrev_b32_e32 v30, 3, v1\n\ts_xor_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v46, v30\n\ts_or_b64 exec, exec, s[0:1]\n\tds_write_b64 v1, v[2:3]\n")
        assert len(chk.scan(bad)) == 1
        # the same for the short form of the else entry (s_andn2_saveexec as the block's first instruction)
        with open(bad, "w") as fh:
            fh.write("_Zk:\n\tv_lshlrev_b32_e32 v15, 3, v1\n\tds_write_b64 v15, v[2:3]\n.LBB0_1:\n\ts_andn2_saveexec_b64 s[0:1], s[6:7]\n"
                     "\tv_mov_b32_e32 v15, v88\n\ts_or_b64 exec, exec, s[0:1]\n\tds_write_b64 v1, v[2:3]\n")
        assert chk.scan(bad) == []
        with open(bad, "w") as fh:
            fh.write("_Zk:\n\tv_lshlrev_b32_e32 v14, 3, v1\n\tds_write_b64 v14, v[2:3]\n.LBB0_1:\n\ts_andn2_saveexec_b64 s[0:1], s[6:7]\n"
                     "\tv_mov_b32_e32 v15, v88\n\ts_or_b64 exec, exec, s[0:1]\n\tds_write_b64 v1, v[2:3]\n")
        assert len(chk.scan(bad)) == 1
    finally:
        os.remove(bad)


def test_role_divergent_barrier_counts_are_checked(tmp_path):
    """tools/check_kernel_resources.py: the role paths of a kernel (relay4's four parts of a KKT solve, the two sides of the tail kernel's
    loop) sit in wavefront-divergent control flow with the workgroup's barriers inside; the build fails unless every path of a group
    passes the same number of s_barrier instructions (xN on a BEGIN marker: a loop body that runs N times per pass)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_kernel_resources as ckr
    good = tmp_path / "good.s"
    good.write_text("_Zk:\n\t; LPVMPC_ROLE_BEGIN trip x2\n\tv_mov_b32_e32 v1, v2\n\ts_barrier\n\t; LPVMPC_ROLE_END trip\n"
                    "\t; LPVMPC_ROLE_BEGIN trip\n\ts_barrier\n\tv_mov_b32_e32 v1, v2\n\ts_barrier\n\t; LPVMPC_ROLE_END trip\n.Lfunc_end0:\n")
    assert ckr.role_barriers(str(good)) == {"_Zk": {"trip": [2, 2]}}
    bad = tmp_path / "bad.s"
    bad.write_text("_Zk:\n\t; LPVMPC_ROLE_BEGIN relay\n\ts_barrier\n\ts_barrier\n\t; LPVMPC_ROLE_END relay\n"
                   "\t; LPVMPC_ROLE_BEGIN relay\n\ts_barrier\n\t; LPVMPC_ROLE_END relay\n.Lfunc_end0:\n")
    assert ckr.role_barriers(str(bad)) == {"_Zk": {"relay": [2, 1]}}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernel_resources.py"), str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "ROLES" in r.stdout
    # the kept assembly of the build: every group of every kernel balanced
    import glob
    for f in glob.glob(os.path.join(ROOT, "autonomous-racing-lpv-mpp-mpc_amd", "csrc", "build", "admm_solve-*-gfx950.s")):
        for kname, groups in ckr.role_barriers(f).items():
            for grp, counts in groups.items():
                assert len(set(counts)) == 1 and counts[0] > 0, (kname, grp, counts)


def test_hot_region_scan_follows_the_control_flow(tmp_path):
    """tools/check_kernel_resources.py: the per-iteration code is what can EXECUTE between the HOT markers -- a block of the iteration
    that the compiler laid out behind the loop (reached by a branch from inside the region) is counted and checked for scratch accesses,
    a block of other code that merely sits between the markers in the text is not."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_kernel_resources as ckr
    f = tmp_path / "k.s"
    f.write_text("_Zk:\n\tv_mov_b32_e32 v1, v2\n\t; LPVMPC_HOT_BEGIN\n\tv_add_f64 v[2:3], v[2:3], v[4:5]\n\ts_cbranch_scc1 .LBB0_9\n"
                 ".LBB0_2:\n\tv_mul_f64 v[2:3], v[2:3], v[4:5]\n\ts_branch .LBB0_3\n"
                 ".LBB0_7:\n\tscratch_load_dwordx2 v[8:9], off, off\n\ts_endpgm\n"                 # other code, laid out in between: not reachable from the region
                 ".LBB0_3:\n\t; LPVMPC_HOT_END\n\ts_endpgm\n"
                 ".LBB0_9:\n\tscratch_store_dwordx2 off, v[2:3], off\n\ts_branch .LBB0_2\n"      # a block of the iteration behind the loop
                 ".Lfunc_end0:\n")
    h = ckr.hot_path(str(f))["_Zk"]
    assert h == dict(instructions=6, scratch=1, scratch_stores=1, spill_lane_moves=0), h

