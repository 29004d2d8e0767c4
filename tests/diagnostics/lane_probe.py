"""Diagnostic: one controller batch through the long-runner lane, step by step, with the statuses after every phase
(python tests/diagnostics/lane_probe.py [seed] [promote_after] [budget] [calls])."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lpvmpc
from lpvmpc import workloads
from test_gpu_deferral import _dev_call, _host


class Raw:
    def __init__(self, p):
        self.cuda_stream = p


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
promote = int(sys.argv[2]) if len(sys.argv) > 2 else 200
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 100
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B = 1024
w = workloads.controller_batch(B, N=20, seed=seed)
plain = workloads.make_solver(w); plain.reserve(B)
_, o = _dev_call(torch, plain, w, B, False); torch.cuda.synchronize(); ref = _host(o); plain.close()
print("plain: max iters", ref["iters"].max(), "beyond 100:", int((ref["iters"] > 100).sum()), "beyond", promote, ":", int((ref["iters"] > promote).sum()), flush=True)
lane = lpvmpc.Lane(device=0, reserved_cus=8, step_streams=2, ring_entries=16)
print("lane created", flush=True)
eng = workloads.make_solver(w); eng.reserve(B)
eng.set_option("defer_after", 100); eng.set_option("defer_budget", budget)
eng.attach_lane(lane, promote_after=promote)
st = Raw(lane.step_streams[0])
keep = []
for c in range(calls):
    keep.append(_dev_call(torch, eng, w, B, False, stream=st))
    print("call", c, "enqueued", flush=True)
    torch.cuda.synchronize()
    h = _host(keep[-1][1])
    print("call", c, "after sync: pending", int((h["status"] == -11).sum()), "promoted so far", lane.promoted_count(), flush=True)
if os.environ.get("LANE_DEBUG"):
    import ctypes as C
    buf = (C.c_ulonglong * 20)()
    lane._lib.lpvmpc_lane_debug_words.argtypes = [C.c_void_p, C.c_void_p]
    lane._lib.lpvmpc_lane_debug_words(lane._l, buf)
    print("debug: entry cfg word %x expected %x entry %d seen %d" % (buf[0], buf[1], buf[2], buf[3]))
    print("scalars:", np.frombuffer(bytes(buf), dtype=np.float64)[4:11], [hex(v) for v in list(buf)[12:20]], flush=True)
eng.join(st.cuda_stream); torch.cuda.synchronize()
print("joined; drains", eng.lane_drain_count(), "promoted", lane.promoted_count(), flush=True)
for _, o in keep:
    h = _host(o)
    bad = np.nonzero((h["status"] != ref["status"]) | (h["iters"] != ref["iters"]) | (h["polish"] != ref["polish"]))[0]
    print("decisions that differ:", len(bad), "max |du|", float(np.nanmax(np.abs(h["uPred"] - ref["uPred"]))), flush=True)
eng.close(); lane.close()
print("done", flush=True)
