"""Benchmark of the bundle-adjustment hot path on 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shape venice]

One step = one full Levenberg-Marquardt solve of the synthetic Venice-shape problem
(1 778 cameras / 993 923 points / 5 001 946 observations, reference tolerances ftol=1e-4,
xtol=1e-12) from the same start, with every input already resident in HBM.  Reported value:
observations x LM iterations per second over the K timed steps (whole job, max over ranks),
plus LM iterations/s, the HBM-roofline fraction of the dominant kernel (HIP-event timed) and
SciPy's least_squares on the host cores on a bounded sample of the same workload.
With --gpus N > 1 and no WORLD_SIZE in the environment this process touches no GPU: it starts N
rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each) and rank 0
prints the line; under torch.distributed.run the ranks are already there and --gpus must equal
WORLD_SIZE.  Ranks shard the points (strong scaling of the fixed problem) and exchange per-camera
partial sums through RCCL inside libsfmba.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, GB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X vector fp64 (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
# fp64 operations of k_schur_pairs<Pinhole> per observation pair, counted in the kernel's ISA (inner
# loop: 92 v_mul_f64 + 30 v_add_f64 + 100 fused multiply-adds counted twice + 2 v_rcp_f64)
PAIR_FLOP_PINHOLE = 92 + 30 + 2 * 100 + 2


def algorithmic_bytes(n_obs, n_pts, n_cam, n_chunks, n_pairs, n_pair_chunks, cam_dim=6):
    """Compulsory HBM bytes per launch of the per-launch kernels (int32 indices, f64 data).

    pairs    explicit Schur assembly k_schur_pairs: the point index of every pair (4), every point
             (32) and its inverse block (48) once, per chunk its record (16) and one 36-double
             partial row (288; 81 doubles for the 9-number camera)
    bsr      k_pcg_spmv_bsr: the block-CSR matrix once (288 + 4 per block, n_pairs here = blocks),
             three vectors read and two written per camera (240); k_pcg_fused: see main()
    spmv_pm  implicit matvec, point-major half: camera index + seed record per observation (36),
             point index (4), inverse block (48) and product written (32) per point, iterate +
             rotation per camera (120)
    spmv_cm  implicit matvec, camera-major half: point index + seeds per observation (28),
             product gathered once per point (32), camera (96), 48 + 12 per chunk"""
    return {
        "pairs": 4 * n_pairs + 80 * n_pts + (16 + 8 * cam_dim * cam_dim) * n_pair_chunks,
        "spmv_pm": 40 * n_obs + 80 * n_pts + 120 * n_cam,
        "spmv_cm": 28 * n_obs + 32 * n_pts + 96 * n_cam + 60 * n_chunks,
    }


def cpu_baseline(scale, seed, camera_model="pinhole"):
    """scipy.optimize.least_squares with the reference's kwargs over the NumPy restatement of
    fun() on a scaled-down instance of the same synthetic workload."""
    from structure_from_motion_amd import synthetic as syn
    if camera_model == "bal":
        from oracle import bal_oracle as b
        nc, npts, no = syn.SHAPES["venice"]
        nc, npts = max(4, round(nc * scale)), max(8, round(npts * scale))
        p = syn.make_bal_problem(nc, npts, min(max(round(no * scale), 2 * npts), 12 * npts), seed=seed)
        r = b.solve_reference(p)
    else:
        from oracle import ba_oracle as o
        p = syn.make_shape("venice", seed=seed, scale=scale)
        r = o.solve_reference(p, use_lil=False)
    iters = max(r["nfev"] - 1, 1)
    return {
        "value": p["n_obs"] * iters / r["seconds"], "unit": "obs*iter/s", "cores": 1, "kind": "port",
        "sample": (f"venice-shape scaled by {scale}: {p['n_cameras']} cameras / {p['n_points']} points / "
                   f"{p['n_obs']} observations; scipy {__import__('scipy').__version__} least_squares(trf, "
                   f"jac_sparsity, x_scale='jac', ftol=1e-4, xtol=1e-12) over the NumPy fun(); "
                   f"{iters} iterations in {r['seconds']:.1f} s, cost {r['initial_cost']:.4e} -> {r['cost']:.4e}; "
                   f"single-threaded TRF/LSMR/finite differences, host has {os.cpu_count()} CPUs"),
        "lm_iters_per_sec": iters / r["seconds"],
    }


def through_class(shape, seed):
    """The hot path the way the reference's driver runs it (main.py:103-124): after every
    integrated view a new BundleAdjustment(wpSet, K, dist, views) and optimize() on everything seen
    so far, then the write-back of poses and points.  Timed whole - packing the pandas table,
    loading the problem, the solve, reading back, write-back - on a problem of the given shape that
    grows view by view (two observations per point); one untimed loop first, so that handles and
    allocations are warm as they are from the second image on."""
    import contextlib
    import io
    import tempfile
    import pandas as pd
    from structure_from_motion_amd import bundle_adjustment as ba
    from structure_from_motion_amd import synthetic as syn
    p = syn.make_shape(shape, seed=seed)
    dist = np.zeros((1, 5))
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())            # optimize() writes optimize_data.npz
    calls = []
    try:
        for timed in (False, True):
            views, steps, K = syn.growing_tables(p)
            wp = syn.SyntheticWorldPoints(np.empty((0, 3)), pd.DataFrame({"ViewId": [], "FeatureIndex": []}))
            for k, (rows, pts) in enumerate(steps, start=1):
                wp.correspondences = pd.concat((wp.correspondences, rows), ignore_index=True)
                wp.world_points = np.vstack((wp.world_points, pts))
                if not len(wp.correspondences):
                    continue
                with contextlib.redirect_stdout(io.StringIO()):
                    t0 = time.perf_counter()
                    b = ba.BundleAdjustment(wp, K, dist, views[:k + 1])
                    b.optimize()
                    b.write_back()
                    total = time.perf_counter() - t0
                if timed:
                    calls.append(dict(views=k + 1, points=b.n_points, seconds=total, lm_iterations=b.result["iterations"],
                                      stages={k2: round(v, 6) for k2, v in b.timings.items()}))
    finally:
        os.chdir(cwd)
        ba.release_engines()
    last = calls[-1]
    return {"optimize_class_seconds": last["seconds"], "last_call": last,
            "loop_seconds": sum(c["seconds"] for c in calls), "calls": len(calls),
            "seconds_per_call": [round(c["seconds"], 6) for c in calls],
            "what": "BundleAdjustment(...).optimize() + write_back() per integrated view, whole call on the host clock "
                    "(pandas packing, optimize_data.npz, set-up, solve, read-back, write-back)"}


def spawn_ranks(n):
    """Parent of an N-rank run: start the ranks as fresh child processes before anything here has
    touched a GPU (counting devices does not), pass rank 0's line through, fail if any rank fails."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} asked for but this machine shows {have} GPU(s); "
                         f"not falling back to fewer\n")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="venice")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-scale", type=float, default=0.12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--linear-solver", type=int, default=0, help="0 auto, 1 PCG, 2 dense Cholesky")
    ap.add_argument("--schur-mode", type=int, default=0, help="0 auto, 1 implicit, 2 explicit")
    ap.add_argument("--pcg-mode", type=int, default=0, help="0 auto, 1 two launches per iteration, 2 single launch")
    ap.add_argument("--pcg-coarse", type=int, default=0,
                    help="two-level PCG preconditioner: 0 default policy, -1 off, k clusters of k cameras")
    ap.add_argument("--matrix-split", type=int, default=0,
                    help="N > 1: 0/1 reduced matrix replicated on every rank, 2 split by points (a collective per PCG iteration)")
    ap.add_argument("--through-class", action="store_true",
                    help="also time BundleAdjustment.optimize() whole, view by view, on a growing problem of --shape "
                         "(fountain or temple: two observations per point), reported as optimize_class_seconds")
    ap.add_argument("--camera-model", default="pinhole", choices=("pinhole", "bal"),
                    help="pinhole: the 12-number camera of bundle_adjustment.py (the headline workload); "
                         "bal: the 9-number camera of ba_test.py on the same scene")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            spawn_ranks(args.gpus)          # does not return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: "
                         f"launch as many ranks as --gpus names\n")
        sys.exit(2)

    import torch
    from structure_from_motion_amd import synthetic as syn
    from structure_from_motion_amd.distributed import init_engine_comm
    from structure_from_motion_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)

    bal = args.camera_model == "bal"
    if bal:
        nc, npts, no = syn.SHAPES[args.shape]
        if args.scale != 1.0:
            nc, npts = max(4, round(nc * args.scale)), max(8, round(npts * args.scale))
            no = min(max(round(no * args.scale), 2 * npts), min(12, nc) * npts)
        p = syn.make_bal_problem(nc, npts, no, seed=args.seed)
        p["K"] = None
    else:
        p = syn.make_shape(args.shape, seed=args.seed, scale=args.scale)
    if world > 1 and args.matrix_split == 2:
        # (a rank's share of a split matrix is a band of block rows when the points follow the cameras)
        from structure_from_motion_amd.distributed import number_points_by_camera
        p, _ = number_points_by_camera(p)
    solver_opts = dict(linear_solver=args.linear_solver, schur_mode=args.schur_mode, pcg_mode=args.pcg_mode, pcg_coarse=args.pcg_coarse,
                       camera_model=args.camera_model, matrix_split=args.matrix_split)
    eng = Engine(device=local_rank, rank=rank, nranks=world, **solver_opts)
    t0 = time.time()
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"],
                    p["points_2d"], p["K"])
    setup_s = time.time() - t0
    # the same problem loaded again into the same handle (its device memory is reused, as in a loop
    # that adjusts after every image), from host arrays and from arrays already in device memory
    # (three times, the median counts: the second call of a process still pays one-off costs of the
    # runtime, 10 ms against 6.7 ms from the third call on)
    warm = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"],
                        p["points_2d"], p["K"])
        warm.append(time.perf_counter() - t0)
    setup_warm_s = sorted(warm)[1]
    dev_in = [torch.from_numpy(np.ascontiguousarray(p[k])).cuda() for k in ("camera_indices", "point_indices", "points_2d")]
    torch.cuda.synchronize()
    t0 = time.time()
    eng.set_problem_device(p["camera_params"], p["points_3d"], *dev_in, p["K"])
    setup_dev_s = time.time() - t0
    del dev_in
    if world > 1:
        init_engine_comm(eng, rank, world)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        eng.reset_params()
        return eng.solve()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    iters = sum(s.iterations for s in stats)
    n_obs = p["n_obs"]
    last = stats[-1]

    # per-kernel timing from one more solve of the same engine with HIP events around every launch
    # (same stream; on every rank, since the collectives of a solve need all of them).  PCG
    # launches queued after convergence return at once; their (few) microseconds stay in the
    # totals and the per-launch average of the PCG kernels is taken over the iterations that did
    # work, so it errs on the slow side.
    os.environ["SFMBA_PROFILE"] = "1"
    eng.reset_params()
    ps = eng.solve()
    os.environ["SFMBA_PROFILE"] = "0"
    # this rank's partition (the whole problem on one GPU)
    from structure_from_motion_amd.distributed import point_partition
    lo, hi = point_partition(p["point_indices"], p["n_points"], rank, world) if world > 1 else (0, p["n_points"])
    mine = (p["point_indices"] >= lo) & (p["point_indices"] < hi)
    my_obs, my_pts = int(mine.sum()), hi - lo
    chunk = int(os.environ.get("SFMBA_CHUNK", "512"))
    cam_counts = np.bincount(p["camera_indices"][mine], minlength=p["n_cameras"])
    n_chunks = int(np.sum((cam_counts + chunk - 1) // chunk))
    ab = algorithmic_bytes(my_obs, my_pts, p["n_cameras"], n_chunks, ps.schur_pairs, ps.schur_chunks,
                           cam_dim=9 if bal else 6)
    if ps.pcg_single_launch:
        # single-launch PCG: the matrix lives in registers; one iteration moves the 18-word
        # record of every gathered column and of every row written (through memory, past L2)
        # (pipelined kernel: the w record instead, one word per tangent number)
        rec = 8 * (9 if bal else 6) if ps.pcg_pipelined else 144
        ab["bsr"] = rec * ps.schur_blocks + rec * p["n_cameras"]
    else:
        cd = 9 if bal else 6
        ab["bsr"] = (8 * cd * cd + 4) * ps.schur_blocks + 40 * cd * p["n_cameras"]
    families = {}
    for fam in ("linearize", "damp", "schur", "pairs", "bsr", "spmv_pm", "spmv_cm", "pcg_vec", "back", "cost", "dense", "comm"):
        n = getattr(ps, "n_" + fam)
        ms = getattr(ps, "ms_" + fam)
        eff = int(ps.pcg_iterations) if fam in ("bsr", "spmv_pm", "spmv_cm") and n else int(n)
        families[fam] = {"launches": int(n), "effective_launches": eff, "total_ms": ms,
                         "avg_us": 1e3 * ms / eff if eff else 0.0}
    if world == 1:
        del families["comm"]
    single = [f for f in ("pairs", "bsr", "spmv_pm", "spmv_cm") if families[f]["launches"]]
    # the kernel that streams the observations inside the linear solve: the explicit Schur
    # assembly, or (implicit mode) the point-major half of the matvec
    dom = "pairs" if ps.schur_mode == 2 else "spmv_pm"

    def price(f):
        avg_s = families[f]["avg_us"] * 1e-6
        return ab[f] / avg_s / 1e9 if avg_s > 0 else 0.0

    kernel_name = {"pairs": "k_schur_pairs", "spmv_pm": "k_spmv_pm", "spmv_cm": "k_spmv_cm",
                   "bsr": (("k_pcg_pipe / k_pcg_pipe2 (per PCG iteration)" if ps.pcg_coarse_solves else "k_pcg_pipe (per PCG iteration)")
                           if ps.pcg_pipelined else "k_pcg_fused (per PCG iteration)")
                   if ps.pcg_single_launch else "k_pcg_spmv_bsr"}
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
    # (profiles/run_profile.sh -> profiles/traffic.json); null when the workload differs
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and args.shape == "venice" and args.scale == 1.0 and not bal and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from summarize import kernel_sources_digest
        with open(tpath) as f:
            record = json.load(f)
        if record.get("kernel_sources_sha256") != kernel_sources_digest():
            # the counters were taken from other kernels than the ones measured here
            traffic_note = "profiles/traffic.json was recorded from other kernel sources: traffic withheld"
        else:
            for name, rec in record.get("kernels", {}).items():     # template arguments follow the name: k_schur_pairs<Pinhole>
                if name.split("<")[0] == kernel_name[dom]:
                    traffic = rec.get("hbm_bytes_per_launch")
    roof = {"bound": "hbm", "kernel": kernel_name[dom], "achieved": price(dom), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": price(dom) / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": ab[dom], "avg_launch_us": families[dom]["avg_us"],
            "launches_per_solve": families[dom]["effective_launches"],
            "other_kernels": {kernel_name[f]: {"algorithmic_bytes_per_launch": ab[f],
                                               "avg_launch_us": families[f]["avg_us"], "achieved": price(f)}
                              for f in single if f != dom},
            "schur_mode": "explicit" if ps.schur_mode == 2 else "implicit",
            "schur_pairs": int(ps.schur_pairs), "schur_blocks": int(ps.schur_blocks)}
    if traffic_note:
        roof["traffic_note"] = traffic_note
    if dom == "pairs" and not bal and families[dom]["avg_us"] > 0:
        # the pair kernel streams little (one index per pair, every point once) and computes much: its
        # other roof is the vector unit
        tf = ps.schur_pairs * PAIR_FLOP_PINHOLE / (families[dom]["avg_us"] * 1e-6) / 1e12
        roof["vector_fp64"] = {"flop_per_pair": PAIR_FLOP_PINHOLE, "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS}
    multi = None
    if world > 1:
        # rank 0's partition and kernels; what does not shrink with the rank count is the PCG on the
        # replicated reduced system and the collectives
        roof["rank"] = 0
        roof["partition"] = {"points": my_pts, "observations": my_obs}
        multi = {"collectives_per_solve": families["comm"]["launches"],
                 "collective_ms_per_solve": families["comm"]["total_ms"],
                 "replicated_pcg_ms_per_solve": families["bsr"]["total_ms"] + families["pcg_vec"]["total_ms"]
                 if ps.schur_mode == 2 else None,
                 "profiled_solve_ms": 1e3 * ps.solve_seconds,
                 "reduced_matrix": "split by points" if args.matrix_split == 2 and ps.schur_mode == 2 and args.linear_solver != 2 else "replicated"}
        if ps.schur_mode == 2:
            # modelled bytes every rank hands to a collective (before the factor 2 (N - 1) / N of a
            # ring all-reduce): the default form - all upper blocks all-reduced once per LM iteration,
            # PCG replicated, no collective inside it - and the split form (options.matrix_split = 2),
            # in which every rank keeps the blocks of its own points and the product q = S p is
            # summed over the ranks in every PCG iteration
            cd, cv = (9, 54) if bal else (6, 27)
            n_c = int(p["n_cameras"])
            blocks = (int(ps.schur_blocks) + n_c) // 2      # upper blocks (schur_blocks counts both orientations)
            lm, pcg = max(1, ps.iterations), ps.pcg_iterations
            small = 8 * (cv * n_c + world) + 8 * 16
            multi["model"] = {
                "replicated_matrix (default)": {"collectives_per_lm_iteration": 3,
                                              "bytes_per_lm_iteration": small + 8 * (cd * cd * blocks + cv * n_c),
                                              "collectives_per_pcg_iteration": 0},
                "matrix_split_by_points (--matrix-split 2)": {"collectives_per_lm_iteration": 3,
                                                       "bytes_per_lm_iteration": small + 8 * (cd * cd + cv) * n_c,
                                                       "collectives_per_pcg_iteration": 1,
                                                       "bytes_per_pcg_iteration": 8 * cd * n_c,
                                                       "pcg_iterations_per_lm_iteration": pcg / lm}}

    eng.close()
    if dist is not None:
        dist.destroy_process_group()

    klass = None
    if rank == 0 and args.through_class:
        if world > 1 or bal or args.shape not in ("fountain", "temple") or args.scale != 1.0:
            sys.stderr.write("bench.py: --through-class needs one GPU, the pinhole camera and --shape fountain or temple\n")
            sys.exit(2)
        klass = through_class(args.shape, args.seed)

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_scale, args.seed, args.camera_model)

    if rank == 0:
        out = {
            "metric": "BA observations/sec (observations x LM iterations / solve time)",
            "value": n_obs * iters / elapsed, "unit": "obs*iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{args.shape}-shape synthetic problem, "
                                    + ("9-number camera of ba_test.py" if bal else "pinhole camera of bundle_adjustment.py")
                                    + f" (seed {args.seed}"
                                    + (f", scale {args.scale}" if args.scale != 1.0 else "") + "): "
                                    f"{p['n_cameras']} cameras / {p['n_points']} points / {n_obs} observations; "
                                    "one step = one full LM solve (Schur complement + block-Jacobi PCG), "
                                    "ftol=1e-4 xtol=1e-12 gtol=1e-8"),
                       "parallelism": f"point-partitioned x{world}" if world > 1 else "single GPU",
                       "linear_solver": {1: "pcg", 2: "dense-cholesky"}[last.linear_solver],
                       "schur_mode": {1: "implicit", 2: "explicit"}[last.schur_mode],
                       "pcg": "single launch" if last.pcg_single_launch else "launch per iteration",
                       "pcg_preconditioner": (f"block-Jacobi + coarse space of {last.pcg_coarse_dim} in {last.pcg_coarse_solves} "
                                              f"of {last.iterations} solves") if last.pcg_coarse_solves else "block-Jacobi"},
            "lm_iters_per_sec": iters / elapsed,
            "lm_iterations_per_solve": iters / args.steps,
            "pcg_iterations_per_solve": sum(s.pcg_iterations for s in stats) / args.steps,
            "solve_seconds": elapsed / args.steps,
            "initial_cost": last.initial_cost, "final_cost": last.final_cost, "status": last.status,
            "setup_seconds": setup_s,
            "setup_seconds_warm_handle": setup_warm_s,
            "setup_seconds_warm_handle_calls": warm,
            "setup_seconds_device_inputs": setup_dev_s,
            "optimize_seconds_end_to_end": setup_warm_s + elapsed / args.steps,
            "pcg_single_fallbacks_per_solve": sum(s.pcg_single_fallbacks for s in stats) / args.steps,
            # kernel time of the linear solves of the event-bracketed solve per PCG iteration it took:
            # the product launches (or the single launches) plus the vector and coarse-space kernels
            "pcg_us_per_iteration": (1e3 * (families["bsr"]["total_ms"] + families["pcg_vec"]["total_ms"]
                                            + families["spmv_pm"]["total_ms"] + families["spmv_cm"]["total_ms"])
                                     / ps.pcg_iterations) if ps.pcg_iterations else None,
            "roofline": roof, "kernels": families,
            "kernels_note": "per-family totals of one extra solve with HIP events around every launch: launches are "
                            "serialised there (the coarse set-up of the second stream no longer overlaps), so their sum "
                            "exceeds ms_per_step",
            "cpu_baseline": cpu,
        }
        if klass is not None:
            out["optimize_class_seconds"] = klass["optimize_class_seconds"]
            out["optimize_class"] = klass
        if multi is not None:
            out["multi_gpu"] = multi
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
