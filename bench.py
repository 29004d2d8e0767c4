#!/usr/bin/env python3
"""bench.py -- headline benchmark: MPC solves/sec (N=20, nx=6, nu=2) on N MI355X.

A "step" = one pass of the hot path (LPV evaluation + roll-out, QP build, ADMM solve to OSQP's
eps = 1e-3, polish, write-back) over ONE batch of BASELINE.json configs[1]: 1024 LPV-MPC controller
instances, N=20, random x0 along the oval (seeded synthetic inputs), already resident
in HBM.  With --gpus N every rank (one process per GPU) runs its own 1024-instance batch (weak scaling,
independent instances, no data-path collective; torch.distributed over RCCL only for the barrier and the
final max-time / iteration-count reduction).

--workload cfg3 / cfg5 measure configs[2] (planner batch) and configs[4] (planner + controller + plant cascade) with the
same contract; they are extra measurements, the headline is the default.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      algorithmic bytes of the ADMM solve kernel / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline  the CPU oracle port (oracle/lpv_ref.c + oracle/osqp_ref.c) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BATCH = 1024              # instances per GPU (configs[1])
HORIZON = 20


def algorithmic_bytes(iters, N=HORIZON, nx=6, nu=2, w=8, m_rows=None):
    """bytes_iter = w (F + 3 n_z + 4 m); per solve add w (2F + N(nx^2+nx nu) + in + out)."""
    nb = nx + nu
    n_z = (N + 1) * nx + N * nu
    m = m_rows if m_rows is not None else (N + 1) * nx + 6 * N    # controller: 126 equality + 120 inequality rows
    F = (N + 1) * (nb * (nb + 1) // 2 + nb * nb)
    bytes_iter = w * (F + 3 * n_z + 4 * m)
    n_in = nx + N * nu + 2 * (N + 1) + nu
    n_out = n_z + 4
    per_solve = w * (2 * F + N * (nx * nx + nx * nu) + n_in + n_out)
    return float(iters.sum()) * bytes_iter + iters.size * per_solve, bytes_iter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="instances per GPU (default: configs[1] = 1024)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the K steps are issued on round-robin (independent batches overlap, so the few "
                         "slow instances of one batch do not leave the GPU idle); 1 = strictly back-to-back steps")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 = configs[1] (the headline, default); cfg3 = configs[2]: planner LPV-MPP, N=30, L-shape "
                         "(use --batch 4096); cfg4 = configs[3]: --batch (default 65536 / gpus) instances per GPU, half controller, half "
                         "planner, N=20; cfg5 = configs[4]: planner + controller + plant cascade, a step is one 30 Hz "
                         "controller tick of --batch vehicles per GPU (default 8192 / gpus) -- extra measurements, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="headline workload on N > 1 GPUs: weak = every rank runs --steps steps of its own batches (the default, what the "
                         "driver's scaling run measures); strong = the --steps steps (batches j mod 32, the same seeds whatever N) are cut "
                         "contiguously over the ranks with lpvmpc.distributed.shard_range, value = batch x steps / time")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short legs of configs[2], [3] and [4] that the one-GPU headline run appends under config.other_workloads")
    ap.add_argument("--defer", type=int, default=-1,
                    help="straggler deferral (lpvmpc_set_option defer_after): an instance still unsolved after this many ADMM "
                         "iterations is parked and continued, --defer-budget iterations at a time, by the resume pass that follows "
                         "every step on the same stream, so the few many-thousand-iteration instances do not hold the step's stream; "
                         "the timed region ends with the passes that finish everything (lpvmpc_join); 0 = off (every launch lasts as "
                         "long as its slowest instance); default: 100 for the headline workload, 0 otherwise")
    ap.add_argument("--defer-budget", type=int, default=100, help="iterations per resume pass of the straggler deferral (0 = to completion)")
    ap.add_argument("--defer-pool", type=int, default=0, help="entries of each of the two pools of parked instances (0 = max(64, B / 8))")
    ap.add_argument("--fleet-groups", type=int, default=4, help="cfg5: independent sub-fleets (engine pairs) the vehicles are cut into")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs outside the timed region (serial steps, batch latencies, one launch of all distinct "
                         "instances, single-solve latency): what the profiling passes use")
    ap.add_argument("--kernel-variant", type=int, default=0,
                    help="diagnostic: lpvmpc_set_option(kernel_variant) on every engine (0 = default; 3 = the DPP two-wavefront kernel; 9 = the four-wavefront latency form of the controller kernel)")
    ap.add_argument("--dump-results", default="",
                    help="rank 0 writes the gathered (u0, status, iters) of the closing all-gather to this .npz (tests compare them with "
                         "single-GPU solves of the same instances)")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the launcher, the sharding and the collectives with the gloo backend and no device work "
                         "(CPU test of the N > 1 path; the printed line carries \"dry_run\": true and no measurement)")
    args = ap.parse_args()
    if args.defer < 0:
        args.defer = 100 if args.workload == "cfg2" else 0
    if args.streams <= 0:
        args.streams = 4 if args.defer > 0 else 64          # with deferral no launch is long: a few streams fill the GPU

    # --gpus N given to a plain `python bench.py`: become the launcher.  Nothing has touched the GPU yet (torch is not
    # even imported), the children are fresh processes (never an exec of a process that initialised HIP).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if args.dry_run:
        return dry_run(args)

    # independent batches are pipelined over several HIP streams; give the runtime as many hardware queues
    # (without deferral: 16 hardware queues; with 8, the streams that share a queue with a launch holding a many-thousand-iteration instance wait
    # behind it; beyond ~20 the queues are time-sliced and every launch slows down -- tools/queue_sweep.sh)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if args.defer > 0 else str(max(4, min(16, args.streams))))
    import numpy as np
    import torch                      # first: liblpvmpc then binds to the HIP runtime torch loaded
    import torch.distributed as dist
    import lpvmpc
    from lpvmpc import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let --gpus spawn them)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # The CPU baseline (oracle port on the host cores, rank 0 only) runs BEFORE the process group exists: it needs no device, and
    # the other ranks then wait for the rendezvous instead of spinning in an RCCL barrier beside sixteen busy host threads.
    cpu = early_cpu_baseline(args, world) if (rank == 0 and not args.no_cpu_baseline) else None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if args.workload in ("cfg5", "cfg4"):
        try:
            out = (bench_cascade if args.workload == "cfg5" else bench_mixed)(args, rank, local_rank, world, dev)
            if rank == 0:
                if cpu is not None:
                    out["cpu_baseline"] = cpu
                out["config"].update(rank_info(world, local_rank, dev))
                print(json.dumps(out), flush=True)
            else:
                rank_info(world, local_rank, dev)
            return
        finally:
            finish(world)
    planner = args.workload == "cfg3"
    if planner and args.batch == BATCH:
        args.batch = 4096                                  # configs[2]: 4096 planner instances per GPU
    B, N = args.batch, (30 if planner else HORIZON)
    nx = 5 if planner else 6
    # One engine (workspace + output buffers), one HIP stream and one DISTINCT batch per in-flight slot: slot i solves the
    # instances of seed i (+ 1000 rank; rank 0 / slot 0 == seed 0 of SURVEY section 8d), so the pipelined steps never
    # re-solve the same 1024 instances and the rare many-thousand-iteration instances of the distribution are part of the
    # measurement.  Steps issued on different streams are independent.
    S = max(1, args.streams)
    NBAT = max(S, 32)                                   # distinct input batches (seeds) the steps cycle through
    make = workloads.planner_batch if planner else workloads.controller_batch
    strong = args.scaling == "strong" and world > 1
    # strong scaling: ONE sequence of steps (batch j mod 32, the seeds of the one-GPU run) cut contiguously over the ranks
    from lpvmpc.distributed import shard_range
    my_steps = list(range(*shard_range(args.steps, rank, world))) if strong else list(range(args.steps))
    my_warm = list(range(*shard_range(args.warmup, rank, world))) if strong else list(range(args.warmup))
    ws = [make(B, N=N, seed=(1 if planner else 0) + i + (0 if strong else 1000 * rank)) for i in range(NBAT)]
    w = ws[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ins = [dict(x0=t(wi["x0"]), u_prev=t(wi["u_prev"]), curv=t(wi["curv_s"]), u_old=t(wi["u_old"]),
                vel_ref=None if planner else t(wi["vel_ref"]), max_ey=t(wi["max_ey"]) if planner else None) for wi in ws]
    engines = [workloads.make_solver(w, device=local_rank) for _ in range(S)]
    for e in engines:
        e.set_option("kernel_variant", args.kernel_variant)
        e.set_option("defer_pool", args.defer_pool)
        e.set_option("defer_after", args.defer)
        e.set_option("defer_budget", args.defer_budget)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

    def new_outs(n):
        return dict(xPred=torch.empty((n, N + 1, nx), dtype=torch.float64, device=dev),
                    uPred=torch.empty((n, N, 2), dtype=torch.float64, device=dev),
                    status=torch.empty(n, dtype=torch.int32, device=dev), iters=torch.empty(n, dtype=torch.int32, device=dev),
                    resid=torch.empty((n, 4), dtype=torch.float64, device=dev), polish=torch.empty(n, dtype=torch.int32, device=dev))
    for e in engines:
        e.reserve(B)
    # Output sets: a deferred instance writes its results late (lpvmpc.h: the buffers of a deferred call must not be reused while
    # an instance of it is LPVMPC_PENDING).  A max_iter straggler needs ceil(max_iter / budget) resume passes of its engine, i.e.
    # that many steps of that engine = S times as many steps overall: the ring of output sets is that long (a multiple of NBAT).
    if args.defer > 0:
        passes = -(-4000 // args.defer_budget) if args.defer_budget > 0 else 1
        NOUT = NBAT * -(-((passes + 2) * S) // NBAT)
    else:
        NOUT = NBAT
    outs = [new_outs(B) for _ in range(NOUT)]
    out_of = {}                                          # input batch -> output set of its most recent solve
    counter = [0]

    def solve_slot(i, n=B, e=None, o=None):
        """Batch i (inputs i, output set o, default i) on engine / stream e (default i mod S)."""
        e = i % S if e is None else e
        out_of[i] = i if o is None else o
        o, d = outs[out_of[i]], ins[i]
        engines[e].solve_dev(n, d["x0"], d["u_prev"], d["vel_ref"], d["curv"], d["u_old"], d["max_ey"], o["xPred"], o["uPred"],
                             o["status"], o["iters"], o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"],
                             stream=streams[e].cuda_stream)

    def step(jg=None):
        """One step: batch jg mod NBAT (jg = the step's global index; default: this rank's running count) on the next engine."""
        j = counter[0]
        counter[0] += 1
        b = (j if jg is None else jg) % NBAT
        solve_slot(b, e=j % S, o=j % NOUT)
        return b

    def fence():
        if args.defer > 0:                       # finish what the straggler deferral still holds: part of the work being timed
            for i, e in enumerate(engines):
                e.join(streams[i].cuda_stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def sync():
        if args.defer > 0:
            for i, e in enumerate(engines):
                e.join(streams[i].cuda_stream)
        torch.cuda.synchronize()

    # set-up, not a step: a one-instance solve on every stream so that its hardware queue, the kernel's code object and the
    # launch attributes exist before the first step (otherwise each of the S streams pays them inside the timed region when
    # the driver asks for fewer warm-up steps than there are streams)
    for i in range(S):
        solve_slot(i, 1)
    sync()
    for j in my_warm:
        step(j)
    for e in engines:
        e.set_timing(True)            # HIP events on the launch stream around every solve-kernel launch
    fence()
    # device-time marks for config.region_breakdown_ms: the region's start, and -- behind the last step enqueued on each stream -- the end of
    # that stream's steps (main launches and the bounded passes that follow them); what remains of the region is the closing passes
    # of lpvmpc_join, i.e. the tail kernel on whatever is still parked
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_steps = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    ev_start.record(streams[0])
    t0 = time.perf_counter()
    timed_slots = [step(args.warmup + j) for j in my_steps]      # (global index: the warm-up steps come first in the sequence)
    for i in range(S):
        ev_steps[i].record(streams[i])
    fence()
    elapsed = time.perf_counter() - t0
    main_phase_ms = max(ev_start.elapsed_time(ev) for ev in ev_steps)
    k_ms = k_n = 0
    r_ms = r_n = 0
    for e in engines:
        ms_, n_ = e.kernel_time_stats()
        k_ms += ms_; k_n += n_
        if args.defer > 0:
            ms_, n_ = e.resume_time_stats()
            r_ms += ms_; r_n += n_
        e.set_timing(False)
    used = sorted(set(timed_slots))
    it_slot = {i: outs[out_of[i]]["iters"].cpu().numpy().astype(np.int64) for i in used}
    st_slot = {i: outs[out_of[i]]["status"].cpu().numpy() for i in used}
    u0_last = outs[out_of[timed_slots[-1]] if timed_slots else 0]["uPred"][:, 0, :].cpu().numpy()     # (the extras below reuse the output sets)
    m_rows = ((N + 1) * nx + (N + 1) * nx + N * 2) if planner else None
    bytes_slot = {i: algorithmic_bytes(it_slot[i], N=N, nx=nx, m_rows=m_rows)[0] for i in used}
    bytes_iter = algorithmic_bytes(it_slot[used[0]], N=N, nx=nx, m_rows=m_rows)[1]
    bytes_timed = float(sum(bytes_slot[i] for i in timed_slots))            # algorithmic bytes of exactly the K timed launches
    # with straggler deferral a launch on the step's stream runs an instance for at most Kp iterations (the first termination
    # check at or beyond defer_after); the rest of a parked instance's iterations belong to its resume launch
    Kp = -(-args.defer // 25) * 25 if args.defer > 0 else 0
    main_bytes_slot = {i: (algorithmic_bytes(np.minimum(it_slot[i], Kp), N=N, nx=nx, m_rows=m_rows)[0] if Kp else bytes_slot[i]) for i in used}
    main_bytes_timed = float(sum(main_bytes_slot[i] for i in timed_slots))
    iters_timed = float(sum(it_slot[i].sum() for i in timed_slots))
    solved_timed = float(sum((st_slot[i] == 1).sum() for i in timed_slots))

    extras = {}
    if not args.no_extras:
        # round 1's protocol for continuity: the same K steps with EVERY slot solving the seed-0 batch (no instance beyond 700
        # iterations), same streams and engines -- kernel progress separated from the change of protocol
        sync(); t1 = time.perf_counter()
        for j in range(args.steps):
            e = j % S
            o, d = outs[j % NOUT], ins[0]
            engines[e].solve_dev(B, d["x0"], d["u_prev"], d["vel_ref"], d["curv"], d["u_old"], d["max_ey"], o["xPred"], o["uPred"],
                                 o["status"], o["iters"], o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"],
                                 stream=streams[e].cuda_stream)
        sync()
        extras["seed0_replicated_solves_per_s"] = B * args.steps / (time.perf_counter() - t1)
        # strictly serial steps on ONE stream (every step waits for its predecessor's slowest instance), batches of slots 0, 1, ...
        n_serial = min(args.steps, 20)
        sync(); t1 = time.perf_counter()
        for j in range(n_serial):
            solve_slot(j % NBAT, e=0)
        sync()
        extras["single_stream_solves_per_s_per_gpu"] = B * n_serial / (time.perf_counter() - t1)
        # latency of one synchronous batch: every distinct batch once (p50 over batches), and the seed-0 batch alone.  A caller that
        # joins after every call has no use for the bounded pass in between: defer_budget -1 for these legs (the closing pass, i.e. the
        # tail kernel, takes the parked instances straight from the batch's launch)
        if args.defer > 0:
            for e in engines:
                e.set_option("defer_budget", -1)
        lat = []
        for i in range(NBAT):
            sync(); t1 = time.perf_counter(); solve_slot(i); sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        extras["latency_legs_defer_budget"] = -1 if args.defer > 0 else None      # (the timed region above runs with --defer-budget; these legs join after every batch)
        extras["p50_batch_latency_ms"] = float(np.median(lat))
        extras["max_batch_latency_ms"] = float(np.max(lat))
        l0 = []
        for _ in range(7):
            sync(); t1 = time.perf_counter(); solve_slot(0); sync()
            l0.append((time.perf_counter() - t1) * 1e3)
        extras["seed0_batch_latency_ms"] = float(np.median(l0))
        if args.defer > 0:
            for e in engines:
                e.set_option("defer_budget", args.defer_budget)
        # isolated launches of the seed-0 batch (nothing else on the GPU): the per-launch kernel time without neighbours
        engines[0].set_timing(True)
        for _ in range(5):
            solve_slot(0); sync()
        ms_, n_ = engines[0].kernel_time_stats(); engines[0].set_timing(False)
        iso_ms = ms_ / max(n_, 1)
        it0 = outs[0]["iters"].cpu().numpy().astype(np.int64)
        extras["isolated_seed0_launch"] = {"kernel_avg_ms": iso_ms, "launches": n_,
                                           "frac": algorithmic_bytes(np.minimum(it0, Kp) if Kp else it0, N=N, nx=nx, m_rows=m_rows)[0] / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "max_admm_iters": int(it0.max()),
                                           "counts": ("the main launch and the iterations it runs (an instance's first %d)" % Kp) if Kp else "the whole solve"}
        # ONE plain launch (no deferral) of all the distinct instances of this rank (32 x B = 32768 at the defaults): one workspace, one stream
        if not planner and NBAT * B <= 65536:
            cat = lambda k: None if ins[0][k] is None else torch.cat([d[k] for d in ins], dim=0)
            big_in = {k: cat(k) for k in ("x0", "u_prev", "vel_ref", "curv", "u_old", "max_ey")}
            big = workloads.make_solver(w, device=local_rank); big.set_option("kernel_variant", args.kernel_variant)
            big.reserve(NBAT * B); bo = new_outs(NBAT * B)

            def big_launch():
                big.solve_dev(NBAT * B, big_in["x0"], big_in["u_prev"], big_in["vel_ref"], big_in["curv"], big_in["u_old"], big_in["max_ey"],
                              bo["xPred"], bo["uPred"], bo["status"], bo["iters"], bo["resid"], bo["polish"], cf_new=w["cf_new"],
                              lap=w["lap"], stream=streams[0].cuda_stream)
            big_launch(); torch.cuda.synchronize()
            tb = []
            for _ in range(3):
                t1 = time.perf_counter(); big_launch(); torch.cuda.synchronize(); tb.append(time.perf_counter() - t1)
            bi = bo["iters"].cpu().numpy().astype(np.int64)
            extras["single_launch"] = {"instances": NBAT * B, "solves_per_s": NBAT * B / float(np.median(tb)), "ms": float(np.median(tb)) * 1e3,
                                       "max_admm_iters": int(bi.max()),
                                       "frac": algorithmic_bytes(bi, N=N, nx=nx)[0] / float(np.median(tb)) / 1e9 / HBM_PEAK_GBS}
            big.close()
        # p50 latency of ONE solve through the host-array entry point (what a 30 Hz control loop calls once per tick: PCIe
        # copies in and out included)
        one = {k: (v[:1] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k != "track" else v) for k, v in w.items()}
        lat1 = []
        for _ in range(30):
            t1 = time.perf_counter()
            engines[0].solve(one["x0"], one["u_prev"], one["vel_ref"], one["curv_s"], one["u_old"], one["max_ey"], one["cf_new"], one["lap"])
            lat1.append((time.perf_counter() - t1) * 1e3)
        extras["p50_single_solve_latency_ms"] = float(np.median(lat1[5:]))
        # ... and on a handle of its own set to the latency form of the kernel (kernel_variant 9: four wavefronts per instance, what the
        # drop-in controller class selects; the engines above are the throughput configuration: deferral, two wavefronts)
        if not planner and N == 20:
            lat_eng = workloads.make_solver(w, device=local_rank); lat_eng.set_option("kernel_variant", 9); lat_eng.reserve(1)
            lat9 = []
            for _ in range(30):
                t1 = time.perf_counter()
                lat_eng.solve(one["x0"], one["u_prev"], one["vel_ref"], one["curv_s"], one["u_old"], one["max_ey"], one["cf_new"], one["lap"])
                lat9.append((time.perf_counter() - t1) * 1e3)
            lat_eng.close()
            extras["p50_single_solve_latency_ms_latency_form"] = float(np.median(lat9[5:]))
        # ... and of the drop-in classes the ROS nodes call, one vehicle at 30 / 20 Hz (CMAIN:361-363: LPVPrediction + solve per tick;
        # PMAIN: the planner at the launch file's N = 40): host arrays in, attributes out, nothing else on the GPU
        if not planner:
            extras["p50_dropin_tick_ms"] = dropin_tick_ms()
        if not planner:
            extras["lone_instance_iteration_us"] = lone_iteration_us(w, local_rank, dev, args.kernel_variant)

    from lpvmpc.distributed import reduce_stats, gather_results, gather_per_rank
    # every rank's own region time, main phase, largest iteration count and algorithmic bytes (weak scaling: every rank draws its own
    # seeds, i.e. its own 4000-iteration stragglers -- a short timed region ends when the unluckiest rank's slowest instance does)
    per_rank = gather_per_rank([elapsed * 1e3, main_phase_ms, float(max(it_slot[i].max() for i in used)), bytes_timed], device=dev)
    elapsed, agg = reduce_stats(elapsed, [iters_timed, solved_timed, bytes_timed], device=dev)
    # the one collective of the path, after the timed region: first input, status and iteration count of every
    # instance of one batch per rank (B x 4 words per rank over RCCL)
    last = timed_slots[-1] if timed_slots else 0
    info = rank_info(world, local_rank, dev)
    g_u0, g_status, g_iters = gather_results(u0_last, st_slot[last], it_slot[last], B * world, device=dev)
    assert g_u0.shape == (B * world, 2) and g_iters.shape == (B * world,)
    if rank == 0 and args.dump_results:
        np.savez(args.dump_results, u0=g_u0, status=g_status, iters=g_iters, batch=B, world=world,
                 seeds=np.array([(1 if planner else 0) + last + 1000 * r for r in range(world)]))

    if rank == 0:
        total = B * args.steps * (1 if strong else world)
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        bytes_launch = main_bytes_timed / max(len(my_steps), 1)    # mean algorithmic bytes of a timed launch on the step's stream (rank 0)
        achieved = bytes_launch / k_avg_s / 1e9 if k_n else float("nan")
        pmc = load_pmc(B, planner)
        out = {
            "metric": "MPC solves/sec (N=20, nx=6, nu=2)" if not planner else "LPV-MPP planner solves/sec (N=30, nx=5, nu=2)",
            "value": total / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": dict({"workload": ("configs[1]: batch=%d LPV-MPC controller solves per GPU, N=20, random x0 along "
                                         "oval, racing tuning, OSQP defaults + polish, cold start" % B) if not planner else
                                        ("configs[2]: batch=%d LPV-MPP planner solves per GPU (velocity-max cost), N=30, "
                                         "L-shape track, OSQP defaults + polish, cold start" % B),
                            "batch_per_gpu": B, "horizon": N, "nx": nx, "nu": 2, "streams": S, "defer_after": args.defer, "defer_budget": args.defer_budget,
                            "distinct_batches_timed": len(used), "batch_seeds": "step j solves batch j mod %d = seed %d + (j mod %d) + 1000 rank" % (NBAT, 1 if planner else 0, NBAT),
                            "mean_admm_iters": agg[0] / total,
                            "max_admm_iters_rank0": int(max(it_slot[i].max() for i in used)),
                            # an instance is a serial chain of ADMM iterations, so the timed region cannot end sooner than its slowest
                            # instance does: iterations x the time of one iteration of an instance that has a CU to itself (measured in
                            # this run where the extras ran: config.lone_instance_iteration_us; else the figures of DESIGN.md section 5)
                            "slowest_instance_floor_ms": slowest_floor_ms(int(max(it_slot[i].max() for i in used)), planner, args.defer, extras),
                            "timed_region_ms": elapsed * 1e3,
                            # main_phase: until the last stream has finished its steps (main launches + bounded resume passes); tail_only: the rest
                            # of the region, when only the closing passes (the whole-CU tail kernel on the still-parked stragglers) are resident
                            "region_breakdown_ms": {"main_phase": main_phase_ms, "tail_only": max(elapsed * 1e3 - main_phase_ms, 0.0)},
                            # per rank (rank order): the rank's own timed region, its main phase, its largest ADMM iteration count and its
                            # aggregate roofline fraction -- value divides by the MAX region, so a rank that drew a 4000-iteration instance
                            # into a short region sets it; these columns tell that from a scaling loss
                            "per_rank_region_ms": [r[0] for r in per_rank], "per_rank_main_phase_ms": [r[1] for r in per_rank],
                            "per_rank_max_iters": [int(r[2]) for r in per_rank],
                            "per_rank_aggregate_frac": [r[3] / (r[0] * 1e-3) / 1e9 / HBM_PEAK_GBS for r in per_rank],
                            "solved_fraction": agg[1] / total}, **info, **extras),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc.get("traffic_bytes_per_launch"), "traffic_source": pmc.get("traffic_source") or pmc.get("source"),
                         "valu": pmc.get("valu"), "valu_source": pmc.get("source") if pmc.get("valu") else None,
                         # the same counters for the configuration that is timed here (main launches of the deferral mode)
                         "valu_timed": pmc.get("valu_timed"),
                         # the contract bound is HBM; what limits the kernel in practice is vector-instruction issue: its state
                         # lives in LDS and registers, the counter traffic is a hundredth of the algorithmic bytes
                         "limiter": "valu_issue", "limiter_frac": (pmc.get("valu_timed") or {}).get("frac"),
                         "pmc_build": pmc.get("build"), "pmc_matches_build": pmc_matches_build(pmc),
                         "kernel": ("admm_solve_kernel<%d, %d, 4, MFMA sweeps, chains relayed over four wavefronts>" % (nx, N) if (not planner and N == 20 and args.kernel_variant == 9 and args.defer == 0) or planner and ((N == 40 and args.kernel_variant in (0, 8)) or (N == 30 and (args.kernel_variant == 8 or (args.kernel_variant == 0 and (B < 512 or args.defer > 0)))))
                                    else "admm_solve_kernel<%d, %d, 2%s>" % (nx, N, (", equilibration vectors in global memory (3 instances per CU)" if args.kernel_variant == 7 and args.defer == 0
                                                                                     else (", MFMA sweeps" if args.kernel_variant in (0, 4, 6) else ""))
                                                                            if planner else ("" if args.kernel_variant == 3 else ", MFMA sweeps"))),
                         "kernel_avg_ms": k_ms / max(k_n, 1), "launches": k_n,
                         "all_launches_avg_ms": (k_ms + r_ms) / max(k_n + r_n, 1),     # main + resume launches: what a kernel trace averages under the one kernel name
                         "resume_launches": ({"count": r_n, "avg_ms": r_ms / max(r_n, 1),
                                              "note": "straggler deferral: the launches counted above park what is unsolved after %d iterations; "
                                                      "these passes of the same kernel (one behind every step, plus the closing lpvmpc_join passes) "
                                                      "continue the parked instances %d iterations at a time (a few workgroups each).  "
                                                      "algorithmic_bytes_per_launch and frac count only the iterations the main launch runs (an "
                                                      "instance's first %d); aggregate_* count all" % (args.defer, args.defer_budget, Kp),
                                              "algorithmic_bytes_per_batch_all_iterations": bytes_timed / max(len(my_steps), 1)} if args.defer > 0 else None),
                         "algorithmic_bytes_per_launch": bytes_launch, "bytes_per_admm_iteration": bytes_iter,
                         "aggregate_algorithmic_GBps": agg[2] / elapsed / 1e9,
                         "aggregate_frac_per_gpu": bytes_timed / elapsed / 1e9 / HBM_PEAK_GBS,
                         "note": "algorithmic bytes per SURVEY 8(d) (factor + vectors streamed once per ADMM "
                                 "iteration); the kernel keeps them in LDS/registers, so real HBM traffic is far lower. "
                                 "frac is per launch, measured while the launches of the other streams share the GPU "
                                 "(a launch lasts as long as its slowest instance); aggregate_frac_per_gpu = all timed "
                                 "launches' bytes / wall time; config.isolated_seed0_launch is one launch alone on the GPU"},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not planner and not args.no_extras and not args.no_other_workloads:
            # the other single-GPU configurations of BASELINE.json, short legs after the timed region (their own engines and streams)
            for e in engines:
                e.close()
            engines = []
            out["config"]["other_workloads"] = other_workloads(args, rank, local_rank, world, dev)
        print(json.dumps(out), flush=True)

    for e in engines:
        e.close()
    finish(world)


def rank_info(world, local_rank, dev):
    """Who took part: size of the RCCL process group as torch.distributed reports it and every rank's device name (collective:
    called on every rank)."""
    import torch
    import torch.distributed as dist
    name = torch.cuda.get_device_name(local_rank)
    names = [name]
    if world > 1:
        names = [None] * world
        dist.all_gather_object(names, name)
    return {"rccl_world": dist.get_world_size() if world > 1 else 1, "devices": names}


def lib_sha256():
    import hashlib
    try:
        from lpvmpc import _ffi
        return hashlib.sha256(open(_ffi.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None


def pmc_matches_build(pmc):
    """True when the committed counter file was collected with the liblpvmpc.so that is loaded now (tools/pmc_summary.py records
    its sha256), False when it names another build, None when the file carries no build record."""
    b = (pmc or {}).get("build") or {}
    return (b.get("liblpvmpc_sha256") == lib_sha256()) if b.get("liblpvmpc_sha256") else None


def early_cpu_baseline(args, world):
    """cpu_baseline of the selected workload (rank 0, before any process group exists; numpy + the oracle only)."""
    import numpy as np
    from lpvmpc import workloads
    if args.workload == "cfg5":
        import lpvmpc
        c = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))
        return cpu_baseline_cascade(c, lpvmpc.Map("L_shape", 0.2), workloads)
    if args.workload == "cfg4":
        from lpvmpc.distributed import shard_range
        total = args.batch * world if args.batch != BATCH else 65536
        lo, hi = shard_range(total // 2, 0, world)
        ws = [workloads.shard_batch(workloads.controller_batch(total // 2, N=20, seed=2), lo, hi),
              workloads.shard_batch(workloads.planner_batch(total // 2, N=20, seed=2), lo, hi)]
        return cpu_baseline_mixed(ws, hi - lo)
    planner = args.workload == "cfg3"
    B = 4096 if (planner and args.batch == BATCH) else args.batch
    make = workloads.planner_batch if planner else workloads.controller_batch
    return cpu_baseline(make(B, N=30 if planner else HORIZON, seed=1 if planner else 0), planner=planner)


def other_workloads(args, rank, local_rank, world, dev):
    """Short legs of configs[2] (planner N = 30, 32 steps of 4096 instances on 16 streams), configs[3] (12 steps of one GPU's 8192 of the
    mixed batch) and configs[4] (8192 vehicles, 120 controller ticks of the cascade) behind the headline's timed region: value, roofline fraction of the
    dominant kernel, its mean launch time and the iteration means, the same code paths as --workload cfg3 / cfg4 / cfg5."""
    import copy
    res = {}
    legs = (("configs[2]", dict(workload="cfg3", batch=4096, steps=32, warmup=4, streams=16, defer=0)),
            ("configs[3]", dict(workload="cfg4", batch=8192, steps=12, warmup=2, streams=3, defer=0)),
            ("configs[4]", dict(workload="cfg5", batch=8192, steps=120, warmup=4, defer=0)))
    for name, kw in legs:
        a = copy.copy(args)
        for k, v in kw.items():
            setattr(a, k, v)
        a.no_cpu_baseline = True; a.no_extras = True; a.dump_results = ""
        t1 = time.perf_counter()
        try:
            o = {"cfg3": bench_planner_leg, "cfg4": bench_mixed, "cfg5": bench_cascade}[a.workload](a, rank, local_rank, world, dev)
        except Exception as exc:                    # a leg must not take the headline line with it
            res[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            continue
        r = o["roofline"]
        res[name] = {"metric": o["metric"], "value": o["value"], "unit": o["unit"], "steps": o["steps"], "warmup": o["warmup"], "ms_per_step": o["ms_per_step"],
                     "frac": r.get("frac"), "aggregate_frac": r.get("aggregate_frac"), "kernel": r.get("kernel"), "kernel_avg_ms": r.get("kernel_avg_ms"), "launches": r.get("launches"),
                     "workload": o["config"]["workload"], "leg_wall_s": time.perf_counter() - t1}
        for k in ("mean_admm_iters", "mean_admm_iters_controller", "mean_admm_iters_planner", "solved_fraction", "alive_fraction", "real_time_factor",
                  "all_vehicles_ticks_per_s", "planner_mean_admm_iters"):
            if k in o["config"]:
                res[name][k] = o["config"][k]
    return res


def bench_planner_leg(args, rank, local_rank, world, dev):
    """configs[2] in short: args.steps steps of one 4096-instance planner batch (N = 30, L-shape, seed 1) on args.streams streams."""
    import numpy as np
    import torch
    from lpvmpc import workloads
    B, N, nx = args.batch, 30, 5
    w = workloads.planner_batch(B, N=N, seed=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = dict(x0=t(w["x0"]), u_prev=t(w["u_prev"]), curv=t(w["curv_s"]), u_old=t(w["u_old"]), max_ey=t(w["max_ey"]))
    S = max(1, args.streams)
    engines = [workloads.make_solver(w, device=local_rank) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    outs = [dict(xPred=torch.empty((B, N + 1, nx), dtype=torch.float64, device=dev), uPred=torch.empty((B, N, 2), dtype=torch.float64, device=dev),
                 status=torch.empty(B, dtype=torch.int32, device=dev), iters=torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(S)]
    for e in engines:
        e.reserve(B)

    def step(j):
        e, o = engines[j % S], outs[j % S]
        e.solve_dev(B, d["x0"], d["u_prev"], None, d["curv"], d["u_old"], d["max_ey"], o["xPred"], o["uPred"], o["status"], o["iters"], None, None,
                    cf_new=w["cf_new"], lap=w["lap"], stream=streams[j % S].cuda_stream)
    for j in range(args.warmup):
        step(j)
    torch.cuda.synchronize()
    for e in engines:
        e.set_timing(True)
    t0 = time.perf_counter()
    for j in range(args.steps):
        step(j)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms = kn = 0
    for e in engines:
        ms_, n_ = e.kernel_time_stats(); kms += ms_; kn += n_
    it = outs[0]["iters"].cpu().numpy().astype(np.int64); stt = outs[0]["status"].cpu().numpy()
    bl, bi = algorithmic_bytes(it, N=N, nx=nx, m_rows=(N + 1) * nx + (N + 1) * nx + N * 2)
    k_avg_s = kms / max(kn, 1) * 1e-3
    for e in engines:
        e.close()
    return {"metric": "LPV-MPP planner solves/sec (N=30, nx=5, nu=2)", "value": B * args.steps / elapsed, "unit": "solves/s", "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "config": {"workload": "configs[2]: batch=%d LPV-MPP planner solves (velocity-max cost), N=30, L-shape track, OSQP defaults + polish, cold start" % B,
                       "mean_admm_iters": float(it.mean()), "solved_fraction": float((stt == 1).mean())},
            "roofline": {"bound": "hbm", "frac": bl / k_avg_s / 1e9 / HBM_PEAK_GBS, "kernel": ("admm_solve_kernel<5, 30, 2, MFMA sweeps> (plain launches of 512 instances or more; smaller ones take the four-wavefront form of the same arithmetic)" if B >= 512
                                                                                                else "admm_solve_kernel<5, 30, 4, MFMA sweeps, chains relayed over four wavefronts>"),
                         "kernel_avg_ms": kms / max(kn, 1), "launches": kn, "algorithmic_bytes_per_launch": bl, "bytes_per_admm_iteration": bi,
                         # what the chip does: every launch's algorithmic bytes over the leg's wall time (the per-launch figure above divides one
                         # launch's bytes by its duration while the other streams' launches share the chip)
                         "aggregate_frac": bl * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "note": "launches of the %d streams overlap: kernel_avg_ms is a launch sharing the chip" % S}}


def finish(world):
    """End of a rank: nobody tears the process group down while another rank is still busy (rank 0 runs the CPU baseline after
    the timed region and prints last), so every rank waits at one more barrier first."""
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def slowest_floor_ms(max_iters, planner, defer, extras):
    """Lower bound of any timed region that contains an instance of max_iters iterations."""
    lone = extras.get("lone_instance_iteration_us") or {}
    main_us = lone.get("solve_kernel", 3.7 if planner else 2.4)
    if planner or defer <= 0 or "tail_kernel" not in lone:
        return max_iters * main_us * 1e-3
    kp = -(-defer // 25) * 25
    return (min(max_iters, kp) * main_us + max(0, max_iters - kp) * lone["tail_kernel"]) * 1e-3


def dropin_tick_ms(reps=60):
    """p50 wall time of one tick of the drop-in classes (the reference's call sequence: LPVPrediction, then solve), one instance."""
    import numpy as np
    import lpvmpc
    from lpvmpc import workloads
    res = {}
    Q, R, dR = workloads.CTRL_TUNINGS["race"]
    c = lpvmpc.PathFollowingLPV_MPC(Q, R, dR, 20, 1, 1 / 30.0, lpvmpc.Map("oval", 0.2), "OSQP", 0, 0)
    x = np.array([1.5, 0.02, 0.1, 0.03, 3.0, -0.02]); u = np.tile([0.01, 0.2], (20, 1)); vel = np.full(21, 1.5); curv = np.zeros(20)
    tl, ts = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); S, A, B, C = c.LPVPrediction(x, u, vel, curv, 60.0, 1); t1 = time.perf_counter()
        c.solve(x, 0.0, u, False, vel, A, B, C, 10); t2 = time.perf_counter()
        tl.append(t1 - t0); ts.append(t2 - t1)
    k = reps // 6
    res["PathFollowingLPV_MPC_N20"] = {"tick": float(np.median(np.add(tl, ts)[k:]) * 1e3), "LPVPrediction": float(np.median(tl[k:]) * 1e3),
                                       "solve": float(np.median(ts[k:]) * 1e3), "admm_iters": int(c.iters)}
    p = lpvmpc.LPV_MPC_Planner(workloads.PLAN_Q, workloads.PLAN_R, workloads.PLAN_dR, workloads.PLAN_L, 40, 0.05, lpvmpc.Map("L_shape", 0.2), "OSQP")
    x = np.array([2.0, 0.0, 0.0, 0.01, 0.01]); SS = 1.0 + np.arange(41) * 2.0 * 0.05; u = np.tile([0.0, 0.2], (40, 1))
    tl, ts = [], []
    for _ in range(reps // 2):
        t0 = time.perf_counter(); S, A, B, C = p.LPVPrediction(x, SS, u); t1 = time.perf_counter()
        p.solve(x, 0, 0, A, B, C, 2, 0.2); t2 = time.perf_counter()
        tl.append(t1 - t0); ts.append(t2 - t1)
    k = reps // 12
    res["LPV_MPC_Planner_N40"] = {"tick": float(np.median(np.add(tl, ts)[k:]) * 1e3), "LPVPrediction": float(np.median(tl[k:]) * 1e3),
                                  "solve": float(np.median(ts[k:]) * 1e3), "admm_iters": int(p.iters)}
    res["note"] = "ms, p50; one vehicle, host arrays in and attributes out (PCIe both ways), the reference's budget is 33 / 50 ms per tick"
    return res


def lone_iteration_us(w, local_rank, dev, kernel_variant):
    """Time of one ADMM iteration of ONE instance alone on the GPU (nothing else running): the solve kernel, and the whole-CU
    tail kernel that finishes parked stragglers.  Measured by differencing two runs of the same instance with termination
    switched off (max_iter 1100 and 2100; adaptive rho and polish off)."""
    import numpy as np
    import torch
    from lpvmpc import workloads
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a[:1])).to(dev)
    ins = dict(x0=t(w["x0"]), u_prev=t(w["u_prev"]), vel=t(w["vel_ref"]), curv=t(w["curv_s"]), u_old=t(w["u_old"]))
    o = dict(xPred=torch.empty((1, w["N"] + 1, 6), dtype=torch.float64, device=dev), uPred=torch.empty((1, w["N"], 2), dtype=torch.float64, device=dev),
             status=torch.empty(1, dtype=torch.int32, device=dev), iters=torch.empty(1, dtype=torch.int32, device=dev),
             resid=torch.empty((1, 4), dtype=torch.float64, device=dev), polish=torch.empty(1, dtype=torch.int32, device=dev))
    out = {}
    for name, defer in (("solve_kernel", 0), ("tail_kernel", 100)):
        ts = {}
        for mi in (1100, 2100):
            e = workloads.make_solver(w, device=local_rank, max_iter=mi, adaptive_rho=0, polish=0, eps_abs=1e-30, eps_rel=1e-30,
                                      eps_prim_inf=1e-30, eps_dual_inf=1e-30)
            e.set_option("kernel_variant", kernel_variant); e.reserve(1)
            e.set_option("defer_after", defer); e.set_option("defer_budget", 0)
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                e.solve_dev(1, ins["x0"], ins["u_prev"], ins["vel"], ins["curv"], ins["u_old"], None, o["xPred"], o["uPred"], o["status"], o["iters"],
                            o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"], stream=0)
                if defer:
                    e.join(0)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            assert int(o["iters"].cpu()[0]) == mi
            ts[mi] = best
            e.close()
        out[name] = (ts[2100] - ts[1100]) / 1000 * 1e6
    return out


def load_pmc(B, planner):
    """HBM traffic and VALU counters of the solve kernel from the rocprofv3 --pmc passes committed under profiles/ (rocprofv3
    cannot run inside bench.py): they describe one isolated launch of the seed-0 batch, NOT this run -- the file is named in
    the output next to the numbers."""
    if planner or B != BATCH:
        return {}
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if pm.get("batch") == B:
            pm["source"] = "profiles/%s (%s); collected separately, not measured in this run" % (name, pm.get("command", "rocprofv3 --pmc"))
            t = (pm.get("timed") or {}).get("main")
            if t and t.get("valu"):
                # main launches of the timed configuration (deferral 100 / 100): traffic and VALU issue of a launch that parks its
                # stragglers instead of waiting for them
                if t.get("traffic_bytes_per_launch"):
                    pm["traffic_bytes_per_launch"] = t["traffic_bytes_per_launch"]
                    pm["traffic_source"] = "profiles/%s, section timed.main: FETCH_SIZE x 2 + WRITE_SIZE (separate --pmc passes) of the main launches of %s" % (name, pm["timed"].get("command", ""))
                v = t["valu"]
                pm["valu_timed"] = {"issue_cycles": v["issue_cycles"], "frac": v["frac"], "launch_cycles": v["launch_cycles"],
                                    "wave_instr_per_launch": v["wave_instr_per_launch"], "kernel_ns_under_pmc": t.get("kernel_ns_under_pmc"),
                                    "active_share_of_wave_cycles": v.get("active_share_of_wave_cycles"),
                                    "wait_share_of_wave_cycles": v.get("wait_share_of_wave_cycles"),
                                    "source": "profiles/%s, section timed.main (%s): %d main launches, each alone on the GPU while its counters "
                                              "are read" % (name, pm["timed"].get("command", ""), t.get("launches_averaged", 0))}
            pm.pop("timed", None); pm.pop("sq_per_launch", None)
            return pm
    return {}


def launch_ranks(n):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, same command line);
    returns the exit code for the launcher: 0 only if every rank succeeded.  The launcher itself never imports torch."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()        # exactly the PIDs started here
    bad = [c for c in codes if c != 0]
    return 0 if len(codes) == n and not bad else (bad[0] if bad else 1)


def dry_run(args):
    """The multi-rank plumbing of the benchmark without a device: gloo process group, contiguous sharding of ONE global
    batch (configs[3] / configs[4] shapes split it; the headline gives every rank its own 1024), barrier, MAX-over-ranks
    time, SUM of counters and the final (u0, status, iters) all-gather of SURVEY.md section 8e -- on stand-in results."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from lpvmpc.distributed import shard_range, reduce_stats, gather_results, gather_per_rank
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    total = {"cfg2": args.batch * world, "cfg3": args.batch * world, "cfg4": args.batch * world if args.batch != BATCH else 65536, "cfg5": 8192}[args.workload]
    a, b = shard_range(total, rank, world)
    n = b - a
    u0 = np.stack([np.arange(a, b, dtype=np.float64), -np.arange(a, b, dtype=np.float64)], axis=1)      # stand-in results
    status = np.ones(n, dtype=np.int32); iters = (25 * (1 + np.arange(a, b) % 3)).astype(np.int32)
    if world > 1:
        dist.barrier()
    per_rank = gather_per_rank([1.0 * (rank + 1), 0.5 * (rank + 1), float(iters.max()), 0.0])     # (stand-ins: region ms, main phase ms, max iterations)
    elapsed, agg = reduce_stats(1e-3 * (rank + 1), [float(iters.sum()), float(n)])
    g_u0, g_status, g_iters = gather_results(u0, status, iters, total)
    ok = bool(g_u0.shape == (total, 2) and np.array_equal(g_u0[:, 0], np.arange(total)) and int(g_iters.sum()) == int(agg[0])
              and int(agg[1]) == total and g_status.shape == (total,))
    # every rank's shard, gathered: the line shows that the shards tile the global batch
    shards = [[a, b]]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, [a, b])
        shards = box
    if rank == 0:
        strong = args.scaling == "strong" and world > 1 and args.workload == "cfg2"
        print(json.dumps({"metric": "MPC solves/sec (N=20, nx=6, nu=2)", "value": None, "unit": "solves/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": "strong" if strong else "weak",
                          "config": {"workload": args.workload, "global_instances": total, "shard_rank0": [a, b], "shards": shards,
                                     "steps_rank0": list(shard_range(args.steps, 0, world)) if strong else [0, args.steps], "rccl_world": world,
                                     "per_rank_region_ms": [r[0] for r in per_rank], "per_rank_max_iters": [int(r[2]) for r in per_rank],
                                     "max_elapsed_s": elapsed, "gather_ok": ok}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def bench_mixed(args, rank, local_rank, world, dev):
    """configs[3] shape: ONE global batch of 65536 instances (--batch B: B per GPU), half LPV-MPC controller (cfg2 distribution)
    and half LPV-MPP planner (cfg3 distribution on the L-shape track), both N = 20, seed 2, split contiguously over the ranks
    (lpvmpc.distributed.shard_range: every rank builds the same global batch and keeps its slice of each half; no data-path
    collective); a step solves the rank's share (the two halves on two streams), steps are pipelined over --streams stream
    pairs.  value = solves/s over both kinds and all ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from lpvmpc import workloads
    from lpvmpc.distributed import reduce_stats, shard_range, gather_results
    total = args.batch * world if args.batch != BATCH else 65536
    half = total // 2
    lo, hi = shard_range(half, rank, world)
    Bh = hi - lo
    B = 2 * Bh
    ws = [workloads.shard_batch(workloads.controller_batch(half, N=20, seed=2), lo, hi),
          workloads.shard_batch(workloads.planner_batch(half, N=20, seed=2), lo, hi)]
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dev_in = [{k: t(w[k]) for k in ("x0", "u_prev", "vel_ref", "curv_s", "u_old", "max_ey")} for w in ws]
    S = max(1, min(args.streams, 8))                      # 8192 instances per launch fill the GPU: a few pairs in flight are enough
    lanes = []
    for _ in range(S):
        pair = []
        for w, nx in zip(ws, (6, 5)):
            e = workloads.make_solver(w, device=local_rank); e.reserve(Bh)
            if args.defer > 0:          # (extra: straggler deferral on this workload too; both N = 20 kernels have the tail kernel)
                e.set_option("defer_after", args.defer); e.set_option("defer_budget", args.defer_budget); e.set_option("defer_pool", args.defer_pool)
            o = dict(xPred=torch.empty((Bh, 21, nx), dtype=torch.float64, device=dev), uPred=torch.empty((Bh, 20, 2), dtype=torch.float64, device=dev),
                     status=torch.empty(Bh, dtype=torch.int32, device=dev), iters=torch.empty(Bh, dtype=torch.int32, device=dev),
                     resid=torch.empty((Bh, 4), dtype=torch.float64, device=dev), polish=torch.empty(Bh, dtype=torch.int32, device=dev))
            pair.append((e, o, torch.cuda.Stream(device=dev)))
        lanes.append(pair)
    counter = [0]

    def step(n_inst=Bh):
        pair = lanes[counter[0] % S]; counter[0] += 1
        for (e, o, st), d, w in zip(pair, dev_in, ws):
            if args.defer > 0:
                e.join(st.cuda_stream)        # a lane has ONE output set: what an earlier step of this lane left parked is finished before the set is written again (lpvmpc.h)
            e.solve_dev(n_inst, d["x0"], d["u_prev"], d["vel_ref"], d["curv_s"], d["u_old"], d["max_ey"], o["xPred"], o["uPred"], o["status"],
                        o["iters"], o["resid"], o["polish"], cf_new=w["cf_new"], lap=w["lap"], stream=st.cuda_stream)

    def fence():
        if args.defer > 0:
            for pair in lanes:
                for e, _, st in pair:
                    e.join(st.cuda_stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(S):
        step(1)                                           # set-up: queues and code objects (see main)
    counter[0] = 0
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    for pair in lanes:
        for e, _, _ in pair:
            e.set_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kms = [0.0, 0.0]; kn = [0, 0]
    for pair in lanes:
        for i, (e, _, _) in enumerate(pair):
            ms_, n_ = e.kernel_time_stats(); kms[i] += ms_; kn[i] += n_
    its = [lanes[0][i][1]["iters"].cpu().numpy().astype(np.int64) for i in range(2)]
    sts = [lanes[0][i][1]["status"].cpu().numpy() for i in range(2)]
    elapsed, agg = reduce_stats(elapsed, [float(its[0].sum()), float(its[1].sum()), float((sts[0] == 1).sum() + (sts[1] == 1).sum())], device=dev)
    # the one collective, after the timed region: (u0, status, iters) of every instance of the global batch, in instance order
    gathered = [gather_results(lanes[0][i][1]["uPred"][:, 0, :].cpu().numpy(), sts[i], its[i], half, device=dev) for i in range(2)]
    assert all(g[0].shape == (half, 2) for g in gathered)
    out = None
    if rank == 0 and args.dump_results:
        np.savez(args.dump_results, ctrl_u0=gathered[0][0], ctrl_status=gathered[0][1], ctrl_iters=gathered[0][2],
                 plan_u0=gathered[1][0], plan_status=gathered[1][1], plan_iters=gathered[1][2], half=half, world=world, seed=2)
    if rank == 0:
        bl_c, _ = algorithmic_bytes(its[0])
        bl_p, bi_p = algorithmic_bytes(its[1], N=20, nx=5, m_rows=21 * 5 + 21 * 5 + 20 * 2)
        k_avg_s = kms[1] / max(kn[1], 1) * 1e-3
        out = {"metric": "MPC solves/sec (mixed planner + controller, N=20)", "value": 2 * half * args.steps / elapsed, "unit": "solves/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong" if args.batch == BATCH else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "configs[3]: one global batch of %d instances split contiguously over the GPUs (%d on rank 0), half LPV-MPC "
                                      "controller (oval), half LPV-MPP planner (L-shape), N=20, OSQP defaults + polish, cold start" % (total, 2 * Bh),
                          "global_instances": total, "shard_rank0": [lo, hi], "batch_per_gpu": 2 * Bh, "horizon": 20,
                          "mean_admm_iters_controller": agg[0] / half, "mean_admm_iters_planner": agg[1] / half,
                          "solved_fraction": agg[2] / (2 * half), "stream_pairs": S,
                          "controller_kernel_avg_ms": kms[0] / max(kn[0], 1)},
               "roofline": {"bound": "hbm", "achieved": bl_p / k_avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bl_p / k_avg_s / 1e9 / HBM_PEAK_GBS,
                            "traffic": None, "kernel": "admm_solve_kernel<5, 20, 2, MFMA sweeps>", "kernel_avg_ms": kms[1] / max(kn[1], 1), "launches": kn[1],
                            "algorithmic_bytes_per_launch": bl_p, "bytes_per_admm_iteration": bi_p,
                            "aggregate_algorithmic_GBps": (bl_c + bl_p) * args.steps * world / elapsed / 1e9,
                            "aggregate_frac": (bl_c + bl_p) * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,      # per GPU: both halves' launches over the wall time
                            "note": "dominant kernel = the planner half (about 12x the controller's iterations)"}}
    for pair in lanes:
        for e, _, _ in pair:
            e.close()
    return out if rank == 0 else None


def cpu_baseline_mixed(ws, Bh):
    """The oracle's controller and planner ticks on the first 256 instances of each half of the rank-0 share, all usable host cores."""
    import numpy as np
    from oracle import osqp_ref
    cores = usable_cores()
    n = min(256, Bh)
    sub = lambda w: {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == Bh and k != "track" else v) for k, v in w.items()}
    t1 = time.perf_counter(); reps = 0
    while time.perf_counter() - t1 < 10.0:
        osqp_ref.ctrl_tick_batch(sub(ws[0]), nthreads=cores); osqp_ref.plan_tick_batch(sub(ws[1]), nthreads=cores); reps += 1
    tt = time.perf_counter() - t1
    return {"value": 2 * n * reps / tt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d x (%d controller + %d planner instances of the same batch), oracle C ticks under OpenMP, %.1f s" % (reps, n, n, tt)}


def cascade_starts(c, B, seed):
    """Monte-Carlo starts of configs[4]: the reference's state at the lap event (fixture data) with perturbed y, yaw and speed."""
    import numpy as np
    rng = np.random.default_rng(seed)
    plant0 = np.tile(c["plant0"], (B, 1))
    plant0[:, 1] += rng.normal(0, 0.01, B); plant0[:, 6] += rng.normal(0, 0.01, B); plant0[:, 2] += rng.uniform(-0.05, 0.3, B)
    return plant0


def bench_cascade(args, rank, local_rank, world, dev):
    """configs[4] shape: Monte-Carlo fleet, planner (N = 40 @ 20 Hz, the launch file's horizon: N = 30 cannot be handed
    off) + Controller_TT (N = 20 @ 30 Hz) + plant, everything on the device; vehicles are split over
    the ranks, no collective on the data path.  value = ALIVE vehicle-ticks/s: the controller ticks that vehicles entered with a
    finite plant state (a vehicle whose planner QP went primal infeasible carries NaN from then on and costs no iterations);
    the all-vehicles figure of earlier rounds is kept as config.all_vehicles_ticks_per_s."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import lpvmpc
    from lpvmpc import workloads as W
    from lpvmpc.distributed import reduce_stats
    B = args.batch if args.batch != BATCH else max(1, 8192 // world)
    c = np.load(os.path.join(ROOT, "tests", "golden", "cascade.npz"))          # state at the reference's lap event (fixture data)
    mp = lpvmpc.Map("L_shape", 0.2)
    plant0 = cascade_starts(c, B, 3 + 1000 * rank)
    Qr, Rr, dRr = W.CTRL_TUNINGS["race"]
    # The vehicles are independent, so the fleet is cut into G sub-fleets, each with its own planner / controller engine pair and
    # streams: a sub-fleet's planner launch ends in a tail of slow QPs during which most of the chip would idle, and the other
    # sub-fleets' launches fill it.  Every tick of every vehicle is the same computation as with G = 1.
    G = max(1, min(args.fleet_groups, B))
    cut = [B * g // G for g in range(G + 1)]
    fleets = []
    for g in range(G):
        sl = slice(cut[g], cut[g + 1]); n = cut[g + 1] - cut[g]
        plan = lpvmpc.BatchedSolver("planner", 40, 0.05, W.PLAN_Q, W.PLAN_R, W.PLAN_dR, L_cf=W.PLAN_L, track=mp.PointAndTangent, device=local_rank)
        plan.handoff_setup()
        ctrl = lpvmpc.BatchedSolver("controller", 20, 1 / 30.0, Qr, Rr, dRr, track=mp.PointAndTangent, device=local_rank)
        ctrl.cascade_init(plan, plant0[sl], np.tile(c["cmd0"], (n, 1)), np.tile(c["uPred0"], (n, 1, 1)), half_width=mp.halfWidth, slack=mp.slack,
                          plan_max_ey=0.2)
        fleets.append((ctrl, plan))

    def read_all():
        outs = [ct.cascade_read(full=False) for ct, _ in fleets]
        o = {k: np.concatenate([x[k] for x in outs]) for k in ("plant", "lap", "plan_iters")}
        o["ticks"] = outs[0]["ticks"]
        o["alive_ticks"] = np.concatenate([ct.cascade_alive_ticks() for ct, _ in fleets]).astype(np.int64)
        return o

    if args.warmup > 0:
        for ct, _ in fleets:
            ct.cascade_tick(args.warmup)
    for ct, pl in fleets:
        ct.set_timing(True); pl.set_timing(True)
    before = read_all()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for ct, _ in fleets:
        ct.cascade_tick(args.steps)
    o = read_all()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    alive = np.all(np.isfinite(o["plant"]), axis=1)
    alive_vt = float((o["alive_ticks"] - before["alive_ticks"]).sum())        # vehicle-ticks of the timed region that did work
    pst = [pl.kernel_time_stats() for _, pl in fleets]; cst = [ct.kernel_time_stats() for ct, _ in fleets]
    pms, pn = sum(x[0] for x in pst), sum(x[1] for x in pst); cms, cn = sum(x[0] for x in cst), sum(x[1] for x in cst)
    laps_done = o["lap"].astype(np.int64) - 1
    reached = [float(np.sum(laps_done >= k)) for k in range(0, 4)]
    elapsed, agg = reduce_stats(elapsed, [float(alive.sum()), alive_vt] + reached, device=dev)
    out = None
    if rank == 0:
        it = o["plan_iters"].astype(np.int64)
        bytes_launch, bytes_iter = algorithmic_bytes(it, N=40, nx=5, m_rows=41 * 5 + 41 * 5 + 40 * 2)
        bytes_launch /= G                                                       # one launch covers one sub-fleet
        k_avg_s = pms / max(pn, 1) * 1e-3
        fleet = B * world
        out = {"metric": "closed-loop alive vehicle-ticks/sec (planner N=40 @20 Hz + controller N=20 @30 Hz + plant)", "value": agg[1] / elapsed,
               "unit": "vehicle-ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong" if args.batch == BATCH else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "configs[4]: %d vehicles per GPU, planner + controller cascade per 30 Hz tick, L-shape track, Monte-Carlo starts "
                                      "around the lap-event state, cold start" % B, "vehicles_per_gpu": B, "sub_fleets": G,
                          "value_counts": "controller ticks entered with a finite plant state (sum over the fleet / wall time); a lost vehicle costs no iterations and is not counted",
                          "all_vehicles_ticks_per_s": fleet * args.steps / elapsed,
                          "alive_share_of_vehicle_ticks": agg[1] / (fleet * args.steps),
                          "alive_fraction": agg[0] / fleet,
                          "fleet_fraction_that_completed_laps": {str(k): agg[2 + k] / fleet for k in range(4)},
                          "laps_completed_survivors_rank0": ({"min": int(o["lap"][alive].min()) - 1, "p50": float(np.median(o["lap"][alive])) - 1,
                                                              "max": int(o["lap"][alive].max()) - 1} if alive.any() else None),
                          "driving_time_s": (args.steps + args.warmup) / 30.0,
                          "attrition_note": "losses = planner QPs of the reference's open-loop recursion turning primal infeasible; the CPU oracle "
                                            "cascade loses the same share (profiles/r02_cascade_attrition_cpu_oracle.txt, tests/test_gpu_cascade.py)",
                          "real_time_factor": (args.steps / 30.0) / elapsed, "planner_ticks": int(o["ticks"][1]),
                          "planner_mean_admm_iters": float(it[alive].mean()) if alive.any() else None,
                          "ctrl_kernel_avg_ms": cms / max(cn, 1)},
               "roofline": {"bound": "hbm", "achieved": bytes_launch / k_avg_s / 1e9 if pn else float("nan"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": (bytes_launch / k_avg_s / 1e9 / HBM_PEAK_GBS) if pn else float("nan"), "traffic": None,
                            "kernel": "admm_solve_kernel<5, 40, 4, MFMA sweeps, chains relayed over four wavefronts>", "kernel_avg_ms": pms / max(pn, 1), "launches": pn,
                            # the planner launches of the timed region (the last tick's iteration counts stand for all: a thinning fleet makes this an
                            # under-estimate for the earlier ticks) over the wall time; the controller's launches add about a twentieth
                            "aggregate_frac": (bytes_launch * pn / elapsed / 1e9 / HBM_PEAK_GBS) if pn else float("nan"),
                            "note": "planner solve kernel (95 % of a tick); algorithmic bytes from the iteration counts of the last planner tick; the sub-fleets' launches overlap, so kernel_avg_ms is the duration of a launch sharing the chip"}}
    for ct, pl in fleets:
        ct.close(); pl.close()
    return out


def cpu_baseline_cascade(c, mp, W, vehicles=64, ticks=300):
    """The oracle cascade (oracle/cascade_ref.py) on a small fleet with the bench's start distribution, all usable host cores, long
    enough (300 controller ticks = 10 s of driving) to go through the same attrition; value = alive vehicle-ticks/s like the
    device's."""
    import numpy as np
    from oracle import cascade_ref
    cores = usable_cores()
    plant0 = cascade_starts(c, vehicles, 3)
    ref = cascade_ref.CascadeRef(mp.PointAndTangent, W.CTRL_TUNINGS["race"], (W.PLAN_Q, W.PLAN_R, W.PLAN_dR, W.PLAN_L), plant0,
                                 np.tile(c["cmd0"], (vehicles, 1)), np.tile(c["uPred0"], (vehicles, 1, 1)), half_width=mp.halfWidth,
                                 slack=mp.slack, plan_max_ey=0.2, nthreads=cores)
    t0 = time.perf_counter(); done = 0; alive_vt = 0
    for _ in range(ticks):
        alive_vt += int(np.sum(np.all(np.isfinite(ref.plant), axis=1)))
        ref.tick(); done += 1
        if time.perf_counter() - t0 > 40.0:
            break
    t = time.perf_counter() - t0
    alive_end = float(np.mean(np.all(np.isfinite(ref.plant), axis=1)))
    return {"value": alive_vt / t, "unit": "vehicle-ticks/s", "cores": cores, "kind": "port",
            "all_vehicles_ticks_per_s": vehicles * max(done, 1) / t, "alive_fraction_at_end": alive_end,
            "sample": "%d vehicles x %d controller ticks of oracle/cascade_ref.py (the bench's start distribution; C tick functions under OpenMP, "
                      "hand-off and plant in numpy), %.1f s; alive vehicle-ticks counted like the device's" % (vehicles, done, t)}


def cpu_baseline(w, target_s=12.0, planner=False):
    """The CPU oracle port (same tick: LPV roll-out + sparse assembly + OSQP restatement, float64) on a
    bounded sample of the same workload, on all host cores (OpenMP over instances)."""
    import numpy as np
    from oracle import osqp_ref
    tick = osqp_ref.plan_tick_batch if planner else osqp_ref.ctrl_tick_batch
    cores = usable_cores()
    sub = lambda n: {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == w["x0"].shape[0] and k != "track" else v)
                     for k, v in w.items()}
    tick(sub(8), nthreads=cores)                    # warm the thread pool / page in
    t0 = time.perf_counter(); tick(sub(64), nthreads=cores); t64 = time.perf_counter() - t0
    n = int(min(w["x0"].shape[0], max(64, 64 * target_s / max(t64, 1e-6))))
    reps, done, t_all = 0, 0, 0.0
    while t_all < target_s and reps < 5000:
        t0 = time.perf_counter(); tick(sub(n), nthreads=cores); t_all += time.perf_counter() - t0
        done += n; reps += 1
    t1 = time.perf_counter(); tick(sub(64), nthreads=1); t_single = time.perf_counter() - t1
    return {"value": done / t_all, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d x the first %d instances of the same batch (oracle/lpv_ref.c + oracle/osqp_ref.c, OpenMP over "
                      "instances, %.1f s)" % (reps, n, t_all),
            "single_core_solves_per_s": 64 / t_single,
            "python_path": cpu_python_path(w, planner),
            # the reference's own Python cost per tick BEFORE the solver runs -- survey container, one core of a
            # Xeon @ 2.6 GHz, the osqp call stubbed out (the wheel is absent); quoted, not measured here
            "reference_presolver_ms_quoted": {"controller_N20": 2.42, "planner_N30": 3.95, "planner_N40": 5.92,
                                              "source": "BASELINE.md section 2: LPVPrediction + _buildMat* + dense->CSR up to the OSQP call, mean of 200, 1 core"}}


def cpu_python_path(w, planner, ticks=48):
    """the reference-shaped Python path per tick on ONE host core -- the numpy restatement of LPVPrediction and
    of the dense QP assembly (oracle/lpv_ref.py; CTRL:166-258, 382-529 / PLAN:242-320, 145-181, 434-486) and the C restatement of OSQP
    called through ctypes as the reference calls the wheel (dense -> CSC conversion included).  The first `ticks` instances of the batch."""
    import numpy as np
    from oracle import lpv_ref as L, osqp_ref as O
    p = dict(L.DEFAULT_PARAMS)
    N = int(w["N"])
    n = min(ticks, w["x0"].shape[0])
    t_lpv = t_asm = t_sol = 0.0
    its = 0
    for j in range(n):
        t0 = time.perf_counter()
        if planner:
            S, A, Bm = L.plan_lpv_prediction(p, w["dt"], N, w["track"], w["x0"][j], w["curv_s"][j], w["u_prev"][j])
        else:
            S, A, Bm = L.ctrl_lpv_prediction(p, w["dt"], N, w["track"], w["x0"][j], w["u_prev"][j], w["vel_ref"][j],
                                             None if w["curv_s"] is None else w["curv_s"][j], w["cf_new"], w["lap"])
        t1 = time.perf_counter()
        if planner:
            mey = float(np.broadcast_to(w["max_ey"], (w["x0"].shape[0],))[j])
            qp = L.plan_build_qp(w["Q"], w["R"], w["dR"], w["L_cf"], N, A, Bm, w["x0"][j], w["u_old"][j], mey, p["max_vel"], p["min_vel"])
        else:
            qp = L.ctrl_build_qp(w["Q"], w["R"], w["dR"], N, A, Bm, w["x0"][j], w["u_old"][j], w["vel_ref"][j], p["max_vel"])
        t2 = time.perf_counter()
        r = O.solve_qp(qp.P, qp.q, qp.A, qp.l, qp.u)
        t3 = time.perf_counter()
        t_lpv += t1 - t0; t_asm += t2 - t1; t_sol += t3 - t2; its += int(r.info.iter)
    return {"ms_per_tick": (t_lpv + t_asm + t_sol) / n * 1e3, "lpv_prediction_ms": t_lpv / n * 1e3, "qp_assembly_ms": t_asm / n * 1e3,
            "solver_ms": t_sol / n * 1e3, "pre_solver_ms": (t_lpv + t_asm) / n * 1e3, "ticks_per_s": n / (t_lpv + t_asm + t_sol),
            "mean_admm_iters": its / n, "cores": 1, "kind": "port",
            "what": "numpy restatement of LPVPrediction + dense assembly (oracle/lpv_ref.py) + ordering, CSC conversion and the C solver "
                    "through ctypes (oracle/osqp_ref.py), %s N = %d, one tick at a time on one core" % ("planner" if planner else "controller", N),
            "sample": "the first %d instances of the same batch" % n}


if __name__ == "__main__":
    main()
