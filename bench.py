#!/usr/bin/env python3
"""bench.py — assembled DoF/s of the SIP Poisson matrix (p=3, 3-D) on N MI355X.

One "step" = one full pass of the hot path: volume + SIP face/interface terms + CSR value write of the
whole matrix, inputs resident in HBM (what the reference times around assemble_system():
examples/poisson.cc:1099-1103).  Workload = BASELINE.json configs[2]: unit cube, 64^3 hex background
mesh, 32^3 = 32 768 polytopes of 2x2x2 cells, p = 3, QGauss(4) cell and face rules (examples/poisson.cc:
702-709), SIP variant of examples/poisson.cc.  Headline FE is FE_DGQ(3) ((p+1)^3 = 64 dofs/polytope,
2 097 152 dofs); the FE_AggloDGP(3) number (20 dofs/polytope, what poisson.cc instantiates) is reported
under "extra".  N > 1 (one rank per GPU): STRONG scaling by default (BASELINE.json north_star: ">= 6x strong scaling to
8 GPUs") - the N = 1 problem split into N contiguous row ranges of whole polytopes, every rank describing only its own
polytopes and their ghost neighbours; `--scaling weak` stacks N such cubes along z instead (subdivided_hyper_rectangle,
64 x 64 x 64N cells, ONE connected problem, rank r owns slab r).  Either way every rank owns its rows outright (owner-computes-rows): the data path has no collective.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, datasheet; see DESIGN.md
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_work(flat, n, owned=None):
    """flops and compulsory HBM bytes of one pass.
    flops: volume 2 d Nq n^2 per polytope; interior face 24 Nqf n^2 (12 of them in the two diagonal blocks, 12 in the two
    coupling blocks); boundary face 6 Nqf n^2.
    bytes: every CSR value written once (8 n^2 (1 + #nbrs) per polytope) + quadrature data read ONCE: 8 (d+1) per volume point,
    8 (2d+1) per face point - an interior face counts once, not once per side - + the small tables of 8(d) (16 d bbox, 4 per
    sub-cell, 16 per face, 4 per block offset).  `bytes_total` is that figure for the whole pass (what a single-launch
    algorithm - the row kernel - is charged with); `bytes[k]` are the compulsory bytes of the diagonal-block / coupling-block
    launch of the two-kernel algorithms, each of which has to read the face data it works on itself.
    `owned` (rank-local descriptions): mask of the polytopes whose rows this rank writes - a face cut by the partition
    then counts with its owned side only (one diagonal-block contribution, one coupling block; its points read once)."""
    import numpy as np
    c = flat.c
    d = c.dim
    arr = flat.arrays()
    nq = flat.nq_tot
    fq_ptr = arr["fq_ptr"]
    cnt = np.diff(fq_ptr).astype(np.float64) if fq_ptr is not None else np.zeros(0)
    n2 = float(n) * n
    if not c.n_faces:
        sides_pts = bdr_pts = int_pts = 0.0
        n_blocks = 0.0
        n_faces_read = 0.0
    else:
        interior = arr["face_out"] >= 0
        if owned is None:
            sides = np.where(interior, 2.0, 0.0)  # owned sides of every interior face
            bdr = ~interior
        else:
            sides = np.where(interior, owned[arr["face_in"]].astype(np.float64) + owned[np.maximum(arr["face_out"], 0)], 0.0)
            bdr = (~interior) & owned[arr["face_in"]]
        sides_pts = float((cnt * sides).sum())     # face points x owned sides
        int_pts = float(cnt[sides > 0].sum())      # points of the interior faces this rank works on, once each
        bdr_pts = float(cnt[bdr].sum())
        n_blocks = float(sides.sum())              # coupling blocks written
        n_faces_read = float((sides > 0).sum() + bdr.sum())
    n_own = c.n_agg if owned is None else int(owned.sum())
    nq_own = float(nq) if owned is None else float(np.diff(arr["vq_ptr"])[owned].sum())
    fl_diag = 2.0 * d * nq_own * n2 + 6.0 * sides_pts * n2 + 6.0 * bdr_pts * n2
    fl_off = 6.0 * sides_pts * n2
    sub_cells = nq_own / float(max(1, (c.degree + 1) ** d))  # informational: 4 B of sub-cell map per cell (QGauss(p+1) rules)
    tables = (16.0 * d + 4.0) * n_own + 4.0 * sub_cells + 16.0 * n_faces_read + 4.0 * n_blocks
    by_diag = 8.0 * n2 * n_own + 8.0 * (d + 1) * nq_own + 8.0 * (2 * d + 1) * (int_pts + bdr_pts) + tables
    by_off = 8.0 * n2 * n_blocks + 8.0 * (2 * d + 1) * int_pts
    by_total = 8.0 * n2 * (n_own + n_blocks) + 8.0 * (d + 1) * nq_own + 8.0 * (2 * d + 1) * (int_pts + bdr_pts) + tables
    return dict(flops=[fl_diag, fl_off], bytes=[by_diag, by_off], bytes_total=by_total,
                bytes_parts={"values": 8.0 * n2 * (n_own + n_blocks), "volume_qdata": 8.0 * (d + 1) * nq_own,
                             "face_qdata": 8.0 * (2 * d + 1) * (int_pts + bdr_pts), "tables": tables})


def make_variant(pa, name, fe):
    if name == "diffusion_reaction":
        return pa.SipVariant.diffusion_reaction(fe)
    if name == "assemble_dg_matrix":
        return pa.SipVariant.assemble_dg_matrix()
    return pa.SipVariant.poisson_example(fe)


def build_handler(pa, dim, cells, block, basis, degree, nq, stack=1, grown=False, distort=0.0):
    lg = cells.bit_length() - 1
    if stack > 1:  # `stack` unit cubes on top of each other (last direction), lexicographic cells: slab r = rank r's rows
        grid = pa.BackgroundGrid.subdivided_hyper_rectangle(dim, (cells,) * (dim - 1) + (cells * stack,), (0.0,) * dim,
                                                            (1.0,) * (dim - 1) + (float(stack),))
    elif (1 << lg) == cells:
        grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, lg)
    else:
        grid = pa.BackgroundGrid.subdivided_hyper_cube(dim, cells, 0.0, 1.0)
    if distort:  # interior vertices moved by up to distort * h (GridTools::distort_random stand-in, exact_solutions_dgp.cc:306)
        grid.distort(distort, 3)
    ah = pa.AgglomerationHandler(grid)
    if grown:  # METIS stand-in: connected irregular agglomerates of about block^dim cells (staircase faces, many neighbours)
        ah.define_grown_agglomerates(block ** dim, seed=1)
    else:
        ah.define_block_agglomerates(block)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(dim, degree)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    return grid, ah, fe


def run_gpu(pa, torch, dist, args, basis, rank, world, local_rank, steps, warmup, alg="auto", look_for_tensor_rules=True, grown=False,
            distort=0.0, diag_first=True):
    t0 = time.time()
    stack = world if args.scaling == "weak" else 1
    grid, ah, fe = build_handler(pa, args.dim, args.cells, args.block, basis, args.degree, args.degree + 1, stack, grown, distort)
    t_handler = time.time() - t0
    var = make_variant(pa, args.variant, fe)
    n = fe.n_dofs_per_cell
    n_agg = ah.n_agglomerates
    # contiguous dof-row ranges of whole polytopes per rank (weak: slab r of the stacked mesh; strong: 1/N of the cube)
    from polydeal_amd.partition import balanced_row_splits, row_range
    if world > 1 and args.scaling == "strong":
        # ranges of whole polytopes that balance the non-zeros a rank writes (rows x row length), not the polytope count
        splits = balanced_row_splits(ah.blocks_per_row(), n, world)
    else:
        splits = [row_range(n_agg, n, r, world)[0] for r in range(world)] + [n_agg * n]
    r0, r1 = splits[rank], splits[rank + 1]
    if world > 1:
        # every rank describes ONLY its own polytopes + their ghost neighbours (pdh_problem.local = 1), like an MPI rank of
        # the reference (source/agglomeration_handler.cc:1026-1091)
        flat = ah.flatten_local(var, r0, r1, diag_first=diag_first, with_colind=False, row_splits=splits)
    else:
        flat = ah.flatten(var, diag_first=diag_first, with_colind=False)
    if not look_for_tensor_rules:  # general-point paths only (pdh_problem::vq_tensor_n / fq_tensor_n < 0)
        flat.c.vq_tensor_n = flat.c.fq_tensor_n = -1
    t_flatten = time.time() - t0 - t_handler
    ctx = pa.Context(local_rank)  # first one in the process: HIP runtime + device context creation (not problem set-up)
    t_context = time.time() - t0 - t_handler - t_flatten
    ctx.set_algorithm(alg)
    # The library overlaps its two kernels on large problems (two streams, ~3 % faster).  The timed region runs them one
    # after the other so that the per-kernel HIP-event durations the roofline uses are those of kernels that have the
    # device to themselves (and match rocprofv3); the overlapped step time is measured afterwards and reported beside it.
    ctx.set_overlap(False)
    ctx.set_problem(flat, r0, r1)
    alg_used = ctx.algorithm_in_use()
    rows_kernel = ctx.rows_kernel_in_use()
    t_setup = time.time() - t0
    setup_parts = {"handler_s": t_handler, "flatten_s": t_flatten, "context_s": t_context,
                   "set_problem_s": t_setup - t_handler - t_flatten - t_context}
    t_setup -= t_context
    for _ in range(warmup):
        ctx.assemble_device()
    ctx.synchronize()
    ctx.set_profiling(True)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t1 = time.perf_counter()
    for _ in range(steps):
        ctx.assemble_device()
    sync_all()
    dt = time.perf_counter() - t1
    kms, nl = ctx.kernel_times_ms()
    ctx.set_profiling(False)
    dt_overlap = None
    if args.overlap_extra:
        ctx.set_overlap(True)
        ctx.assemble_device()
        sync_all()
        t2 = time.perf_counter()
        for _ in range(steps):
            ctx.assemble_device()
        sync_all()
        dt_overlap = time.perf_counter() - t2
        ctx.set_overlap(False)
    per_rank = None
    if world > 1:
        # every rank's own wall time and kernel time (HIP events): the first real multi-GPU run must show WHICH rank is slow
        dev_t = "cpu" if args.rehearse_on_one_gpu else "cuda"
        mine = torch.tensor([dt / steps * 1e3, kms[0] + kms[1], float(ctx.stats()["n_owned_agg"]), float(ctx.stats()["n_values"])],
                            dtype=torch.float64, device=dev_t)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows_r = [[float(v) for v in a.cpu()] for a in allr]
        kmax, kmin = max(r[1] for r in rows_r), min(r[1] for r in rows_r)
        per_rank = {"wall_ms_per_step": [r[0] for r in rows_r], "kernel_ms": [r[1] for r in rows_r],
                    "polytopes": [int(r[2]) for r in rows_r], "nnz": [int(r[3]) for r in rows_r],
                    "kernel_ms_max_over_min": kmax / max(kmin, 1e-9)}
        t = torch.tensor([dt], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = ctx.stats()
    nnz = int(stats["n_values"])  # values of the rows this rank owns; the matrix has the sum over ranks
    if world > 1:
        tn = torch.tensor([float(nnz)], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(tn)
        nnz = int(round(float(tn.item())))
    mfma = ctx.kernel_work()
    own_mask = None
    if world > 1:
        dofs = flat.arrays()["dof_offset"]
        own_mask = (dofs >= r0) & (dofs < r1)
    work = algorithmic_work(flat, n, own_mask) if rank == 0 else None
    # Validity of what the timed launches left in HBM, checked on the device (nothing is copied back): no non-finite entry,
    # and - FE_DGQ is a partition of unity - the sum of all entries of the owned rows is 1^T A 1 restricted to them, which
    # the SIP form fixes in closed form: sum over the owned Nitsche boundary faces of sigma |F|  +  c |owned polytopes|
    # (jumps and gradients of a constant vanish).  examples' test analogue: test/polydeal/poisson_sanity_check_01.
    import numpy as np
    cs = ctx.checksum()
    validity = {"non_finite": cs["non_finite"], "max_abs": cs["max_abs"], "sum": cs["sum"]}
    if basis == "dgq":
        arr = flat.arrays()
        exp = var.reaction_c * float(np.sum(arr["vq_w"]))
        if flat.c.n_faces:
            bd = arr["face_out"] < 0
            fw = np.add.reduceat(arr["fq_w"], arr["fq_ptr"][:-1])  # (not differences of a running sum: those lose 1e-11)
            if world > 1:  # local description: only boundary faces of OWNED polytopes belong to these rows
                own = (arr["dof_offset"] >= r0) & (arr["dof_offset"] < r1)
                bd = bd & own[arr["face_in"]]
            exp += float(np.sum(arr["face_sigma"][bd] * fw[bd]))
        validity.update(expected_sum=exp, rel_err=abs(cs["sum"] - exp) / max(abs(exp), 1e-300),
                        note="sum of all entries = 1^T A 1 = sum_bdry sigma |F| + c |Omega| for FE_DGQ")
    vals = ctx.assemble() if stats["n_values"] <= 64_000_000 else None
    chk = validity
    aux = None
    if world == 1 and args.aux_kernels and alg == "auto" and look_for_tensor_rules and not grown:
        try:
            aux = time_aux_kernels(torch, ctx, flat, n, stats)
        except Exception as exc:
            aux = {"error": repr(exc)}
    ctx.close()
    ghost = None
    if world > 1 and args.exchange_extra and alg == "auto":
        try:
            ghost = run_ghost_exchange(pa, torch, dist, args, flat, r0, r1, rank, world, local_rank, steps, warmup, vals)
        except Exception as exc:  # the measured default path above must not be lost to a failure of the extra variant
            ghost = {"error": repr(exc)}
    return dict(n_dofs=ah.n_dofs, n_agg=n_agg, n=n, dt=dt, kms=kms, nl=nl, stats=stats, work=work, mfma=mfma,
                t_setup=t_setup, nnz=nnz, checksum=chk, alg=alg_used, dt_overlap=dt_overlap,
                ghost=ghost, local=world > 1, aux=aux, setup_parts=setup_parts, rows_kernel=rows_kernel, per_rank=per_rank)


def time_aux_kernels(torch, ctx, flat, n, stats):
    """The callers' other device work on the resident problem, through the device-pointer entry points (nothing allocated or
    copied per call): right-hand side (k_rhs; examples/poisson.cc:745-759, 788-828) and evaluation of u_h and grad u_h at the
    volume quadrature points (k_eval; PolyUtils::compute_global_error, include/poly_utils.h:1686-1731).  Both are HBM-bound."""
    import numpy as np
    arr = flat.arrays()
    d = flat.c.dim
    nq, nqf = flat.nq_tot, flat.nqf_tot
    dev = "cuda"
    f_vol = torch.rand(max(nq, 1), dtype=torch.float64, device=dev)
    g_b = torch.rand(max(nqf, 1), dtype=torch.float64, device=dev)
    rhs = torch.zeros(stats["n_owned_agg"] * n, dtype=torch.float64, device=dev)
    sol = torch.rand(stats["n_owned_agg"] * n, dtype=torch.float64, device=dev)
    pts = torch.from_numpy(np.ascontiguousarray(arr["vq_x"])).to(dev)
    ptr = torch.from_numpy(np.ascontiguousarray(arr["vq_ptr"])).to(dev)
    u = torch.zeros(nq, dtype=torch.float64, device=dev)
    g = torch.zeros(d * nq, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    reps = 5

    def timed(fn):
        fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps

    t_rhs = timed(lambda: ctx.assemble_rhs_device(f_vol.data_ptr(), g_b.data_ptr(), rhs.data_ptr()))
    t_ev = timed(lambda: ctx.evaluate_device(sol.data_ptr(), ptr.data_ptr(), pts.data_ptr(), nq, u.data_ptr(), g.data_ptr()))
    # PolyUtils::compute_global_error fused on the device (pdh_global_error_device): evaluation + JxW-weighted sums in one
    # kernel, 16 bytes per polytope come back
    w_d = torch.from_numpy(np.ascontiguousarray(arr["vq_w"])).to(dev)
    eu_d = torch.rand(nq, dtype=torch.float64, device=dev)
    eg_d = torch.rand(d * nq, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t_err = timed(lambda: ctx.global_error_sums_device(sol.data_ptr(), ptr.data_ptr(), pts.data_ptr(), nq, w_d.data_ptr(),
                                                       eu_d.data_ptr(), eg_d.data_ptr()))
    by_err = 8.0 * d * nq + 8.0 * (d + 2) * nq + 8.0 * sol.numel() + 16.0 * stats["n_owned_agg"]
    nbd = float(np.diff(arr["fq_ptr"])[arr["face_out"] < 0].sum()) if flat.c.n_faces else 0.0
    # compulsory bytes of the right-hand side: volume points (x, JxW, f); of the face points only the boundary ones carry a
    # datum (x, n, JxW, sigma, g) - the kernel reads the 8-byte map entry of every packed face point to find them
    by_rhs = (8.0 * (d + 2) * nq + 8.0 * stats["n_face_side_points"] + 8.0 * (2 * d + 3) * nbd + 8.0 * rhs.numel())
    by_ev = 8.0 * d * nq + 8.0 * (d + 1) * nq + 8.0 * sol.numel()
    return {"k_rhs": {"ms": 1e3 * t_rhs, "algorithmic_bytes": by_rhs, "GBs": by_rhs / t_rhs * 1e-9, "frac_of_hbm_peak": by_rhs / t_rhs * 1e-9 / HBM_PEAK_GBS},
            "k_eval": {"ms": 1e3 * t_ev, "points": nq, "algorithmic_bytes": by_ev, "GBs": by_ev / t_ev * 1e-9,
                       "frac_of_hbm_peak": by_ev / t_ev * 1e-9 / HBM_PEAK_GBS, "note": "u_h and grad u_h at the volume quadrature points"},
            "k_eval_error_sums": {"ms": 1e3 * t_err, "points": nq, "algorithmic_bytes": by_err, "GBs": by_err / t_err * 1e-9,
                                  "frac_of_hbm_peak": by_err / t_err * 1e-9 / HBM_PEAK_GBS,
                                  "note": "compute_global_error on the device: u_h, grad u_h and the weighted L2 / H1 sums in one kernel "
                                          "(incl. the 16 B per polytope copied back and the stream synchronisation)"}}


def end_to_end(pa, args):
    """What a caller that assembles ONCE per mesh sees (every example of the reference does): the gather of the quadrature data
    (source/agglomeration_handler.cc:622-707 agglomerated_quadrature, :1103-1243 reinit_master - inside the span the reference
    times, examples/poisson.cc:1099-1106, and inside the CPU baseline's span) + the library's set-up + one assembly, for the headline
    problem.  Two ways to hand the problem over: with its points (flatten: the host mirror gathers 29 M points, the library checks
    and uploads them) and without them (flatten_cartesian + pdh_set_problem_cartesian: groups of points named by cell and local
    face, generated on the device).  The agglomeration handler itself (define_agglomerate, connectivity, dof numbering: set-up in
    the reference too) is timed apart."""
    out = {}
    t0 = time.perf_counter()
    grid, ah, fe = build_handler(pa, args.dim, args.cells, args.block, args.fe, args.degree, args.degree + 1)
    out["handler_s"] = time.perf_counter() - t0
    var = make_variant(pa, args.variant, fe)
    for name in ("points", "cartesian"):
        best = None
        for rep in range(2):
            ctx = pa.Context(0)
            t1 = time.perf_counter()
            flat = ah.flatten(var, True, False) if name == "points" else ah.flatten_cartesian(var, True, False)
            t2 = time.perf_counter()
            ctx.set_problem(flat)
            t3 = time.perf_counter()
            ctx.assemble_device()
            ctx.synchronize()
            t4 = time.perf_counter()
            cs = ctx.checksum()
            rec = {"describe_s": t2 - t1, "set_problem_s": t3 - t2, "first_assembly_s": t4 - t3, "total_s": t4 - t1,
                   "value": ah.n_dofs / (t4 - t1), "rows_kernel": ctx.rows_kernel_in_use(), "checksum_sum": cs["sum"],
                   "non_finite": cs["non_finite"]}
            ctx.close()
            del flat
            if best is None or rec["total_s"] < best["total_s"]:
                best = rec
        out[name] = best
    out["unit"] = "DoF/s = n_dofs / (describe + set_problem + first assembly), best of 2; handler_s not included"
    return out


def strong_proxy(pa, args, steps):
    """Strong-scaling readiness measured on ONE device (last bullet: at ~2 k polytopes per GPU occupancy, not xGMI, is
    the risk): the share of rank 0 in an N-rank strong-scaling run - the first 1/N of the rows, with its ghost neighbours - timed
    for N = 1, 2, 4, 8, on the N = 1 bench problem and on the "~1 M DoF" problem of the north star (16 384 polytopes).  With
    owner-computes-rows there is no exchange step, so t(1) / t(1/N) on one device is what N devices would give, up to load
    imbalance between the ranges (none on these meshes) and the barrier."""
    from polydeal_amd.partition import row_range
    out = {}
    nq = args.degree + 1
    probs = {"bench_problem": None}
    if args.dim == 3:
        probs["about_1M_dofs"] = (args.cells, args.cells, max(args.block, args.cells // 2))
    for name, shape in probs.items():
        if shape is None:
            grid, ah, fe = build_handler(pa, args.dim, args.cells, args.block, args.fe, args.degree, nq)
        else:
            grid = pa.BackgroundGrid.subdivided_hyper_rectangle(args.dim, shape, (0.0,) * args.dim,
                                                                tuple(float(c) / shape[0] for c in shape))
            ah = pa.AgglomerationHandler(grid)
            ah.define_block_agglomerates(args.block)
            fe = (pa.FE_DGQ if args.fe == "dgq" else pa.FE_AggloDGP)(args.dim, args.degree)
            ah.initialize_fe_values(nq, nq)
            ah.distribute_agglomerated_dofs(fe)
        var = make_variant(pa, args.variant, fe)
        n, n_agg = fe.n_dofs_per_cell, ah.n_agglomerates
        res = {"n_polytopes": n_agg, "n_dofs": ah.n_dofs, "ranges": {}}
        t1 = None
        for N in (1, 2, 4, 8):
            r0, r1 = row_range(n_agg, n, 0, N)
            splits = [row_range(n_agg, n, r, N)[0] for r in range(N)] + [n_agg * n]
            flat = ah.flatten(var, True, False) if N == 1 else ah.flatten_local(var, r0, r1, diag_first=True, with_colind=False,
                                                                                 row_splits=splits)
            ctx = pa.Context(0)
            ctx.set_overlap(False)
            ctx.set_problem(flat, r0, r1)
            for _ in range(3):
                ctx.assemble_device()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.assemble_device()
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / steps
            ctx.set_profiling(True)
            for _ in range(steps):
                ctx.assemble_device()
            ctx.synchronize()
            kms, _ = ctx.kernel_times_ms()
            ctx.set_profiling(False)
            alg = ctx.algorithm_in_use()
            ctx.close()
            if N == 1:
                t1 = wall
            res["ranges"]["1/%d" % N] = {"polytopes": (r1 - r0) // n, "ms_per_step": 1e3 * wall, "kernel_ms": kms[0] + kms[1],
                                          "algorithm": alg, "speedup_vs_whole": t1 / wall}
        res["t1_over_t8"] = res["ranges"]["1/1"]["ms_per_step"] / res["ranges"]["1/8"]["ms_per_step"]
        res["kernel_t1_over_t8"] = res["ranges"]["1/1"]["kernel_ms"] / max(res["ranges"]["1/8"]["kernel_ms"], 1e-9)
        out[name] = res
    out["note"] = ("one device, rank 0's share of an N-rank strong-scaling split (first 1/N of the rows, rank-local description with ghosts); "
                   "ms_per_step = back-to-back launches incl. launch overhead, kernel_ms = HIP-event kernel time")
    return out


def run_ghost_exchange(pa, torch, dist, args, flat, r0, r1, rank, world, local_rank, steps, warmup, ref_vals):
    """The same step in the reference's distributed form (include/poly_utils.h:1930-1992, 2134-2194): the owner of a face cut
    by the partition computes M21 / M22 and ships them; transport = one all-to-all-v over RCCL (xGMI) per step."""
    import numpy as np
    on_cpu = args.rehearse_on_one_gpu  # gloo rehearsal: the transport goes through host memory
    ctx = pa.Context(local_rank)
    ctx.set_overlap(False)
    ctx.set_exchange_mode("ghost")
    # one stream for kernels and collective: the library launches on a torch stream, the all-to-all is issued under it
    tstream = torch.cuda.Stream()
    if not on_cpu:
        ctx.set_stream(tstream.cuda_stream)
    ctx.set_problem(flat, r0, r1)
    sc, rc = ctx.exchange_layout(world)
    send = torch.zeros(max(sum(sc), 1), dtype=torch.float64, device="cuda")
    recv = torch.zeros(max(sum(rc), 1), dtype=torch.float64, device="cuda")

    def step():
        ctx.assemble_device()
        ctx.exchange_get_send(send.data_ptr())
        if on_cpu:
            ctx.synchronize()
            hs, hr = send[:sum(sc)].cpu(), torch.zeros(sum(rc), dtype=torch.float64)
            dist.all_to_all_single(hr, hs, output_split_sizes=rc, input_split_sizes=sc)
            recv[:sum(rc)].copy_(hr)
            torch.cuda.synchronize()
        else:
            with torch.cuda.stream(tstream):
                dist.all_to_all_single(recv[:sum(rc)], send[:sum(sc)], output_split_sizes=rc, input_split_sizes=sc)
        ctx.exchange_apply(recv.data_ptr())

    for _ in range(max(1, warmup)):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    err = None
    if ref_vals is not None:
        got = ctx.values()
        err = float(np.max(np.abs(got - ref_vals)) / max(np.max(np.abs(ref_vals)), 1e-300))
    tot = torch.tensor([float(sum(sc))], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
    dist.all_reduce(tot)
    ctx.close()
    return {"ms_per_step": 1e3 * float(t.item()) / steps, "bytes_exchanged_per_step": 8.0 * float(tot.item()),
            "max_rel_diff_vs_owner_computes_rows": err,
            "transport": "gloo through host memory (rehearsal)" if on_cpu else "torch.distributed all_to_all_single (RCCL)"}


def effective_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box gives a
    1-GPU job a share of the host's cores; os.cpu_count() reports all of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(pa, args, basis):
    """Reference-shaped C restatement (oracle/sip_ref.c) timed on the host cores on a bounded sample:
    same FE / rules / variant / block size, fewer polytopes (cost per polytope is size-independent).
    A 1-core calibration run sizes the two samples to about args.cpu_seconds of CPU wall time each."""
    from oracle import sip_ref
    cores = effective_cpus()
    nthreads = max(1, min(cores, sip_ref.max_threads()))

    def run(nb, thr, fast=False):
        cells = nb * args.block
        grid, ah, fe = build_handler(pa, args.dim, cells, args.block, basis, args.degree, args.degree + 1)
        flat = ah.flatten(make_variant(pa, args.variant, fe), diag_first=True, with_colind=True)
        kw = flat.arrays()
        c = flat.c
        kw.update(dim=c.dim, degree=c.degree, basis=c.basis, n_agg=c.n_agg, n_faces=c.n_faces, n_rows=c.n_rows,
                  diag_first=c.diag_first, reaction_c=c.reaction_c)
        _, secs = sip_ref.assemble(kw, nthreads=thr, fast=fast)
        return dict(dofs=ah.n_dofs, secs=secs, threads=thr, n_agg=ah.n_agglomerates, nb=nb)

    def sized(fast):
        cal = run(3, 1, fast)
        per_poly = max(cal["secs"] / cal["n_agg"], 1e-7)
        root = 1.0 / args.dim
        nb_one = int(max(3, min(args.cpu_max_blocks, round((args.cpu_seconds / per_poly) ** root))))
        nb_all = int(max(3, min(args.cpu_max_blocks, round((args.cpu_seconds * nthreads / per_poly) ** root))))
        return dict(all=run(nb_all, nthreads, fast), one=run(nb_one, 1, fast))

    return dict(port=sized(False), fast=sized(True)), cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--cells", type=int, default=64, help="background cells per direction")
    ap.add_argument("--block", type=int, default=2, help="cells per direction in one polytope")
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--fe", choices=["dgq", "dgp"], default="dgq")
    ap.add_argument("--variant", choices=["poisson", "diffusion_reaction", "assemble_dg_matrix"], default="poisson",
                    help="caller variant (penalty / face ownership / reaction term), SURVEY.md 8(a)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: strong (default; BASELINE.json north_star asks for strong scaling to 8 GPUs) = the N = 1 problem "
                         "split into N contiguous row ranges; weak = N cubes stacked along the last direction, one slab per rank")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary FE measurement")
    ap.add_argument("--general-points-extra", action="store_true",
                    help="also time the row kernel's general-point paths (quadrature declared unstructured); off by default: "
                         "it launches the same kernel symbol and would blur the per-kernel averages of a rocprofv3 trace")
    ap.add_argument("--overlap-extra", action="store_true",
                    help="also time the step with the library's default overlapped launch of its two kernels "
                         "(roofline.overlapped_ms_per_step); off by default so that a rocprofv3 trace of the default command "
                         "contains serialised launches only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong-proxy", action="store_true",
                    help="N = 1: add extra.strong_proxy (rank 0's share of 2-, 4-, 8-rank strong-scaling splits timed on this device); off "
                         "by default: it launches the headline kernel on smaller row ranges, which would blur that kernel's average in a "
                         "rocprofv3 --stats summary of the default command")
    ap.add_argument("--no-aux-kernels", dest="aux_kernels", action="store_false",
                    help="skip the timing of the right-hand-side and evaluation kernels (extra.aux_kernels)")
    ap.add_argument("--no-exchange-extra", dest="exchange_extra", action="store_false",
                    help="N>1: skip the second measurement with the reference's ghost-block exchange (M21/M22 shipped over RCCL)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses device 0 and the gloo backend "
                         "(exercises the partitioned code path; the numbers are meaningless)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU wall time of each baseline sample")
    ap.add_argument("--cpu-max-blocks", type=int, default=20, help="cap on polytopes per direction of a CPU sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # Invoked plainly (`python bench.py --gpus N`): start the one-process-per-GPU job as a CHILD before anything here
        # touches the GPU (a process that has initialised HIP must never be replaced by exec), relay its output and exit code.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback for the measured path)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import polydeal_amd as pa

    main_res = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, args.steps, args.warmup)
    extra = {}
    other = "dgp" if args.fe == "dgq" else "dgq"
    if not args.no_extra:
        r2 = run_gpu(pa, torch, dist, args, other, rank, world, local_rank, args.steps, args.warmup)
        extra = {"fe": ("FE_AggloDGP" if other == "dgp" else "FE_DGQ") + "(%d)" % args.degree,
                 "dofs_per_polytope": r2["n"], "n_dofs": r2["n_dofs"],
                 "value": r2["n_dofs"] / (r2["dt"] / args.steps), "ms_per_step": 1e3 * r2["dt"] / args.steps,
                 "algorithm": r2["alg"], "rows_kernel": r2["rows_kernel"],
                 "kernel_ms": {"diagonal_blocks": r2["kms"][0], "coupling_blocks": r2["kms"][1]}}
        if r2.get("work"):
            # one launch writes everything (row kernel): SURVEY 8(d) bytes of the pass; two launches: each reads its own face data
            by2 = r2["work"]["bytes_total"] if r2["alg"] == "rows" else sum(r2["work"]["bytes"])
            extra.update(algorithmic_bytes_per_step=by2, hbm_GBs=by2 / (r2["dt"] / args.steps) * 1e-9,
                         frac_of_hbm_peak=by2 / (r2["dt"] / args.steps) * 1e-9 / HBM_PEAK_GBS,
                         frac_note=("SURVEY 8(d) bytes / time; the term kernels read 1-D rules instead of the point arrays that count "
                                    "charges (more than half of it for FE_AggloDGP(3)): measured traffic in "
                                    "profiles/r04b_hbm_small_elements.txt") if r2["rows_kernel"] == "terms" else None,
                         fp64_bound_ms=1e3 * sum(r2["work"]["flops"]) / (FP64_PEAK_TFLOPS * 1e12))
    if not args.no_extra and world == 1 and main_res.get("rows_kernel") == "terms" and main_res["n"] == 64:
        # the same workload through the kernel that was AUTO's choice for FE_DGQ(3) until round 4 (pdh_rows.h: moment form + MFMA
        # contraction, one wave per polytope), same process, same device
        os.environ["PDH_TERMS_DGQ3"] = "0"
        try:
            r9 = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, args.steps, args.warmup)
            extra["pdh_rows_h"] = {"what": "headline workload with PDH_TERMS_DGQ3=0", "rows_kernel": r9["rows_kernel"],
                                   "ms_per_step": 1e3 * r9["dt"] / args.steps, "kernel_ms": r9["kms"][0],
                                   "frac_of_hbm_peak": r9["work"]["bytes_total"] / (r9["kms"][0] * 1e-3) * 1e-9 / HBM_PEAK_GBS}
        except Exception as e:  # noqa: BLE001
            extra["pdh_rows_h"] = {"error": str(e)[:200]}
        finally:
            del os.environ["PDH_TERMS_DGQ3"]
    if not args.no_extra and world == 1 and args.dim == 3 and args.degree == 3:
        # the lower degrees of the same mesh (streamed kinds of the row kernel; degree 2 is the element of BASELINE configs[1] / [3])
        import copy
        extra["lower_degrees"] = []
        for p_low in (2, 1):
            try:
                a2 = copy.copy(args)
                a2.degree = p_low
                r5 = run_gpu(pa, torch, dist, a2, args.fe, rank, world, local_rank, max(3, args.steps // 2), 1)
                t5 = r5["dt"] / max(3, args.steps // 2)
                by5 = r5["work"]["bytes_total"] if r5["alg"] == "rows" else sum(r5["work"]["bytes"])
                extra["lower_degrees"].append({"fe": ("FE_DGQ" if args.fe == "dgq" else "FE_AggloDGP") + "(%d)" % p_low, "n_dofs": r5["n_dofs"],
                                               "algorithm": r5["alg"], "rows_kernel": r5["rows_kernel"], "ms_per_step": 1e3 * t5, "value": r5["n_dofs"] / t5,
                                               "algorithmic_bytes_per_step": by5, "frac_of_hbm_peak": by5 / t5 * 1e-9 / HBM_PEAK_GBS})
            except Exception as exc:  # a secondary measurement must not cost the main line
                extra["lower_degrees"].append({"degree": p_low, "error": repr(exc)})
    if main_res.get("aux") is not None:
        extra["aux_kernels"] = main_res["aux"]
    if not args.no_extra and world == 1 and args.dim == 3:
        # the same cells agglomerated the way the reference's own callers do it (examples/poisson.cc:543-566: METIS on the cell
        # graph; here regions grown over the graph - METIS is not available offline): irregular connected polytopes of about
        # block^3 cells, neighbours met along several planes ("staircase" faces), 2-3x as many neighbours as a block has
        try:
            r3 = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, max(3, args.steps // 2), 1, grown=True)
            by3 = r3["work"]["bytes_total"] if r3["alg"] == "rows" else sum(r3["work"]["bytes"])
            t3 = r3["dt"] / max(3, args.steps // 2)
            extra["irregular_agglomerates"] = {
                "what": "same %d^3 cells, %d connected agglomerates of ~%d cells grown over the cell graph (METIS stand-in)" % (args.cells, r3["n_agg"], args.block ** 3),
                "n_dofs": r3["n_dofs"], "nnz": r3["nnz"], "algorithm": r3["alg"], "ms_per_step": 1e3 * t3, "value": r3["n_dofs"] / t3,
                "kernel_ms": r3["kms"], "algorithmic_bytes_per_step": by3, "hbm_GBs": by3 / t3 * 1e-9,
                "frac_of_hbm_peak": by3 / t3 * 1e-9 / HBM_PEAK_GBS, "setup_s": r3["t_setup"], "checksum": r3["checksum"]}
            if r3["alg"] == "rows":  # what the same problem costs through the two-kernel moment form (round 2's path for such meshes)
                r4 = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, 3, 1, alg="moment", grown=True)
                extra["irregular_agglomerates"]["moment_form_ms_per_step"] = 1e3 * r4["dt"] / 3
                extra["irregular_agglomerates"]["moment_form_kernel_ms"] = r4["kms"]
            extra["irregular_agglomerates"]["rows_kernel"] = r3["rows_kernel"]
            # the element the reference's callers instantiate on such agglomerates (examples/poisson.cc:413 FE_AggloDGP, :543-566
            # METIS): term kernel (pdh_terms.h), against the two-kernel direct form it replaced there
            r6 = run_gpu(pa, torch, dist, args, other, rank, world, local_rank, max(3, args.steps // 2), 1, grown=True)
            t6 = r6["dt"] / max(3, args.steps // 2)
            by6 = r6["work"]["bytes_total"] if r6["alg"] == "rows" else sum(r6["work"]["bytes"])
            line6 = {"fe": ("FE_AggloDGP" if other == "dgp" else "FE_DGQ") + "(%d)" % args.degree, "n_dofs": r6["n_dofs"], "nnz": r6["nnz"],
                     "algorithm": r6["alg"], "rows_kernel": r6["rows_kernel"], "ms_per_step": 1e3 * t6, "value": r6["n_dofs"] / t6,
                     "algorithmic_bytes_per_step": by6, "frac_of_hbm_peak": by6 / t6 * 1e-9 / HBM_PEAK_GBS, "checksum": r6["checksum"]}
            if r6["alg"] == "rows":
                r7 = run_gpu(pa, torch, dist, args, other, rank, world, local_rank, 3, 1, alg="direct", grown=True)
                line6["direct_form_ms_per_step"] = 1e3 * r7["dt"] / 3
            extra["irregular_agglomerates"]["other_element"] = line6
        except Exception as exc:
            extra["irregular_agglomerates"] = {"error": repr(exc)}
    if not args.no_extra and world == 1 and args.dim == 3:
        # Non-Cartesian cells (the reference's own defaults: examples/3D_piston.cc:396-400, exact_solutions_dgp.cc:306): the headline
        # mesh with every interior vertex moved by up to 0.1 h - faces are no longer planar, quadrature rules no longer tensor
        # rules, so AUTO leaves the row kernels; what it takes instead and what that costs by the same SURVEY 8(d) byte count
        try:
            r8 = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, max(3, args.steps // 2), 1, distort=0.1)
            t8 = r8["dt"] / max(3, args.steps // 2)
            by8 = r8["work"]["bytes_total"]
            extra["distorted"] = {
                "what": "headline mesh, interior vertices moved by up to 0.1 h (seed 3): general hexahedra, non-planar faces",
                "n_dofs": r8["n_dofs"], "nnz": r8["nnz"], "algorithm": r8["alg"], "rows_kernel": r8["rows_kernel"],
                "ms_per_step": 1e3 * t8, "value": r8["n_dofs"] / t8, "kernel_ms": r8["kms"],
                "algorithmic_bytes_per_step": by8, "frac_of_hbm_peak": by8 / t8 * 1e-9 / HBM_PEAK_GBS, "checksum": r8["checksum"]}
            # the same mesh with plain ascending columns (Epetra / Trilinos rows: no shifted blocks, the coupling kernel writes whole lines)
            r8a = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, max(3, args.steps // 2), 1, distort=0.1, diag_first=False)
            t8a = r8a["dt"] / max(3, args.steps // 2)
            extra["distorted"]["ascending_layout"] = {"ms_per_step": 1e3 * t8a, "kernel_ms": r8a["kms"], "algorithm": r8a["alg"],
                                                      "frac_of_hbm_peak": by8 / t8a * 1e-9 / HBM_PEAK_GBS}
        except Exception as exc:
            extra["distorted"] = {"error": repr(exc)}
    if world == 1 and not args.no_extra and args.dim == 3:
        try:
            extra["end_to_end"] = end_to_end(pa, args)
        except Exception as exc:
            extra["end_to_end"] = {"error": repr(exc)}
    if world == 1 and args.strong_proxy and not args.no_extra:
        try:
            extra["strong_proxy"] = strong_proxy(pa, args, max(5, args.steps))
        except Exception as exc:
            extra["strong_proxy"] = {"error": repr(exc)}
    direct = None
    if main_res["alg"] != "direct" and not args.no_extra:
        # the same workload through the direct (MFMA contraction) form, for the record
        direct = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, max(3, args.steps // 3), 1, alg="direct")

    general = None
    if main_res["alg"] == "rows" and args.general_points_extra and world == 1:
        # the same workload with the row kernel's general-point paths (no use of the tensor structure of the rules)
        general = run_gpu(pa, torch, dist, args, args.fe, rank, world, local_rank, max(3, args.steps // 3), 1,
                          look_for_tensor_rules=False)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
      try:  # a failure of the CPU leg must not discard the GPU measurement already taken
        res, cores = cpu_baseline(pa, args, args.fe)
        a, o = res["port"]["all"], res["port"]["one"]
        fa, fo = res["fast"]["all"], res["fast"]["one"]
        cpu = {"value": a["dofs"] / a["secs"], "unit": "DoF/s", "cores": a["threads"], "kind": "port",
               "sample": "oracle/sip_ref.c (reference-shaped C restatement, gcc -O2, OpenMP over polytopes) on %d^%d=%d polytopes "
                         "of the same workload (%d dofs) in %.1f s on %d threads; 1 core: %d polytopes in %.1f s"
                         % (a["nb"], args.dim, a["n_agg"], a["dofs"], a["secs"], a["threads"], o["n_agg"], o["secs"]),
               "value_1core": o["dofs"] / o["secs"], "host_cores": cores,
               "best_effort_cpu": {
                   "note": "same algorithm content, hoisted basis evaluation + vectorised inner loops "
                           "(sipref_assemble_fast, gcc -O3 -march=x86-64-v3): reported so that GPU/CPU is not quoted "
                           "against the reference-shaped port only",
                   "value": fa["dofs"] / fa["secs"], "cores": fa["threads"], "value_1core": fo["dofs"] / fo["secs"],
                   "sample": "%d polytopes in %.1f s on %d threads; 1 core: %d polytopes in %.1f s"
                             % (fa["n_agg"], fa["secs"], fa["threads"], fo["n_agg"], fo["secs"])}}
      except Exception as exc:
        cpu = {"error": repr(exc)}

    if rank == 0:
        r = main_res
        ms_step = 1e3 * r["dt"] / args.steps
        value = r["n_dofs"] / (r["dt"] / args.steps)
        w = r["work"]
        # rank 0's algorithmic work: of the global description scaled to its share, or of its own local description
        frac_rows = 1.0 if r["local"] else r["stats"]["n_owned_agg"] / r["n_agg"]
        # "moment": both kinds of block through pdh_moment.h; "mixed": diagonal blocks moment, coupling blocks direct
        names = ("k_mdiag" if r["alg"] in ("moment", "mixed") else "k_diag", "k_moffdiag" if r["alg"] == "moment" else "k_offdiag")
        if r["n"] > 64:  # blocks in 64 x 64 tiles, one wave per tile (csrc/pdh_tiled.h)
            names = ("k_tdiag", "k_toffdiag")
        moment = r["alg"] != "direct"
        t_k = [max(r["kms"][0] * 1e-3, 1e-9), max(r["kms"][1] * 1e-3, 1e-9)]
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        wl = "%dD cells=%d block=%d %s p=%d" % (args.dim, args.cells, args.block, args.fe, args.degree)
        tj = {}
        traffic_note = None
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath)).get(wl, {})
            except Exception:
                tj = {}
            # PMC byte counts belong to the kernels of ONE library version: a stale entry must not decorate a new kernel
            lib_ver = pa.load_library().pdh_version().decode()
            if tj and tj.get("lib_version") != lib_ver:
                traffic_note = "profiles/traffic.json was collected with %r, this is %r: not reported" % (tj.get("lib_version"), lib_ver)
                tj = {}

        def kernel_entry(i):
            return {"kernel": names[i], "kernel_ms": t_k[i] * 1e3,
                    "algorithmic_flops_per_launch": w["flops"][i] * frac_rows,
                    "algorithmic_bytes_per_launch": w["bytes"][i] * frac_rows,
                    "algorithmic_TFLOPs": w["flops"][i] * frac_rows / t_k[i] * 1e-12,
                    "hbm_achieved_GBs": w["bytes"][i] * frac_rows / t_k[i] * 1e-9,
                    "traffic": tj.get(names[i] + "_bytes")}

        ke = [kernel_entry(0), kernel_entry(1)]
        if r["alg"] == "rows":
            # ONE kernel writes every block of the rows (pdh_rows.h): its algorithmic bytes / flops are those of the whole step
            # (values once, volume q-data once, every face's q-data ONCE)
            tot_b, tot_f = w["bytes_total"] * frac_rows, (w["flops"][0] + w["flops"][1]) * frac_rows
            # the kernel that ran: term kernels (pdh_terms.h; FE_DGQ(3): the workgroup form, pdh_terms_wg.h) or a kind of pdh_rows.h
            kname = ("k_terms_wg" if r["n"] == 64 else "k_terms") if r.get("rows_kernel") == "terms" else "k_rows"
            ke[0] = {"kernel": kname, "kernel_ms": t_k[0] * 1e3, "algorithmic_flops_per_launch": tot_f,
                     "algorithmic_bytes_per_launch": tot_b, "algorithmic_TFLOPs": tot_f / t_k[0] * 1e-12,
                     "hbm_achieved_GBs": tot_b / t_k[0] * 1e-9, "traffic": tj.get(kname + "_bytes")}
        if r["alg"] == "rows":
            roof = {"bound": "hbm", "kernel": ke[0]["kernel"], "achieved": ke[0]["hbm_achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ke[0]["hbm_achieved_GBs"] / HBM_PEAK_GBS, "traffic": ke[0]["traffic"],
                    "kernel_ms": ke[0]["kernel_ms"], "launches_timed": r["nl"],
                    "traffic_GBs": None if ke[0]["traffic"] is None else ke[0]["traffic"] / (ke[0]["kernel_ms"] * 1e-3) * 1e-9,
                    "traffic_note": ("`achieved` is SURVEY 8(d)'s algorithmic byte count / kernel time; the term kernels read the 1-D rules of a "
                                     "polytope from one record gathered on the device at pdh_set_problem (3 KB per polytope here) instead of "
                                     "the point arrays, so measured traffic lies BELOW the algorithmic bytes by most of their quadrature share")
                                    if r.get("rows_kernel") == "terms" else None,
                    "algorithmic_bytes_per_launch": ke[0]["algorithmic_bytes_per_launch"],
                    "algorithmic_bytes_parts": {k: v * frac_rows for k, v in w["bytes_parts"].items()},
                    "algorithmic_flops_per_launch": ke[0]["algorithmic_flops_per_launch"],
                    "algorithm": ("term kernel (pdh_terms.h; FE_DGQ(3): pdh_terms_wg.h, a workgroup of four waves per polytope): the owner "
                                  "writes all blocks of its rows; every entry a short sum of products of three 1-D matrix entries (mass / "
                                  "stiffness per cell, trace / flux per sub-face) held in LDS; cells and sub-faces that form tensor grids are "
                                  "summed over as one with composite 1-D rules; no moments, no MFMA"
                                  if r.get("rows_kernel") == "terms" else
                                  "row kernel (pdh_rows.h): one wave per polytope writes all blocks of its 64 rows as whole 128-byte "
                                  "lines; planar axis-aligned faces -> rank-one face moments, Kronecker form C (x) S of the coupling "
                                  "blocks; volume moments + three-stage contraction of the diagonal block on the f64 MFMA"),
                    "whole_step_GBs": ke[0]["algorithmic_bytes_per_launch"] / (r["dt"] / args.steps) * 1e-9,
                    "whole_step_algorithmic_TFLOPs": ke[0]["algorithmic_flops_per_launch"] / (r["dt"] / args.steps) * 1e-12,
                    "quadrature_structure": "tensor rules per sub-cell / sub-face found on the points by pdh_set_problem "
                                            "(QGauss on a Cartesian background grid): moments in factorised form"}
            if general is not None:
                roof["general_points"] = {
                    "note": "same workload, pdh_problem::vq_tensor_n = fq_tensor_n = -1: quadrature points treated as unstructured "
                            "(moment GEMMs over the points on the f64 MFMA); algorithm " + general["alg"],
                    "ms_per_step": 1e3 * general["dt"] / max(3, args.steps // 3), "kernel_ms": general["kms"][0],
                    "frac": ke[0]["algorithmic_bytes_per_launch"] / max(general["kms"][0] * 1e-3, 1e-9) * 1e-9 / HBM_PEAK_GBS}
        elif moment:
            # The moment form removes ~85 % of the arithmetic of SURVEY 8(d)'s count, so the f64-MFMA roof no longer binds:
            # the roof that remains is the HBM traffic of the values (written once) + quadrature data (read once).
            dom = 0 if t_k[0] >= t_k[1] else 1
            roof = {"bound": "hbm", "kernel": names[dom], "achieved": ke[dom]["hbm_achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ke[dom]["hbm_achieved_GBs"] / HBM_PEAK_GBS, "traffic": ke[dom]["traffic"],
                    "kernel_ms": ke[dom]["kernel_ms"], "launches_timed": r["nl"],
                    "algorithmic_bytes_per_launch": ke[dom]["algorithmic_bytes_per_launch"],
                    "algorithmic_flops_per_launch": ke[dom]["algorithmic_flops_per_launch"],
                    "algorithm": "moment form (pdh_moment.h): Legendre moments of the quadrature on the f64 MFMA + sum "
                                 "factorisation; VALU / LDS bound, not at the HBM roof yet",
                    "other_kernel": ke[1 - dom],
                    "overlapped_ms_per_step": None if r["dt_overlap"] is None else 1e3 * r["dt_overlap"] / args.steps,
                    "overlap_note": "library default: the two kernels run concurrently on two streams (pdh_set_overlap); the timed "
                                    "region above serialises them so that kernel_ms are undisturbed per-kernel durations",
                    "whole_step_GBs": w["bytes_total"] * frac_rows / (r["dt"] / args.steps) * 1e-9,
                    "whole_step_algorithmic_TFLOPs": (w["flops"][0] + w["flops"][1]) * frac_rows / (r["dt"] / args.steps) * 1e-12}
            if names[dom] in ("k_diag", "k_offdiag", "k_tdiag", "k_toffdiag"):
                # mixed form whose dominant kernel is a DIRECT (MFMA contraction) one: that kernel is bound by the f64 MFMA
                roof.update(bound="mfma", achieved=ke[dom]["algorithmic_TFLOPs"], peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=ke[dom]["algorithmic_TFLOPs"] / FP64_PEAK_TFLOPS, hbm_achieved_GBs=ke[dom]["hbm_achieved_GBs"])
        if moment and direct is not None:
            if True:
                d = direct
                dk = [d["kms"][0] * 1e-3, d["kms"][1] * 1e-3]
                roof["direct_form"] = {
                    "note": "same workload, pdh_set_algorithm(PDH_ALG_DIRECT): contraction over the points on the f64 MFMA "
                            "(k_diag / k_offdiag); algorithmic flops count both triangles and all four interface blocks, the "
                            "kernels execute 0.48x of them (symmetry, M21 = M12^T)",
                    "ms_per_step": 1e3 * d["dt"] / max(3, args.steps // 3), "k_diag_ms": dk[0] * 1e3, "k_offdiag_ms": dk[1] * 1e3,
                    "k_diag_algorithmic_TFLOPs": w["flops"][0] * frac_rows / dk[0] * 1e-12,
                    "k_diag_executed_mfma_TFLOPs": 512.0 * d["mfma"][0] / dk[0] * 1e-12,
                    "k_diag_executed_frac_of_peak": 512.0 * d["mfma"][0] / dk[0] * 1e-12 / FP64_PEAK_TFLOPS,
                    "fp64_peak_TFLOPs": FP64_PEAK_TFLOPS}
        if not moment:
            ach_tf = ke[0]["algorithmic_TFLOPs"]
            roof = {"bound": "mfma", "kernel": names[0], "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": ke[0]["traffic"], "kernel_ms": ke[0]["kernel_ms"],
                    "launches_timed": r["nl"], "algorithmic_flops_per_launch": ke[0]["algorithmic_flops_per_launch"],
                    "algorithmic_bytes_per_launch": ke[0]["algorithmic_bytes_per_launch"],
                    "hbm_achieved_GBs": ke[0]["hbm_achieved_GBs"], "hbm_peak_GBs": HBM_PEAK_GBS, "k_offdiag": ke[1],
                    "whole_step_TFLOPs": (w["flops"][0] + w["flops"][1]) * frac_rows / (r["dt"] / args.steps) * 1e-12,
                    "executed": {
                        "note": "algorithmic flops (SURVEY 8(d)) count both triangles of the symmetric diagonal blocks and "
                                "all four interface blocks; the kernels compute upper tile pairs only and write "
                                "A[Q,P] = A[P,Q]^T, so `frac` can exceed 1; these are the MFMA flops actually issued "
                                "(v_mfma_f64_4x4x4_4b_f64, 512 flop; measured instruction ceiling 73 TFLOP/s)",
                        "k_diag_flops_per_launch": 512.0 * r["mfma"][0], "k_diag_TFLOPs": 512.0 * r["mfma"][0] / t_k[0] * 1e-12,
                        "k_diag_frac_of_peak": 512.0 * r["mfma"][0] / t_k[0] * 1e-12 / FP64_PEAK_TFLOPS,
                        "k_offdiag_TFLOPs": 512.0 * r["mfma"][1] / t_k[1] * 1e-12}}
        out = {
            "metric": "assembled DoF/s (SIP Poisson, p=%d, %dD)" % (args.degree, args.dim),
            "value": value, "unit": "DoF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dD SIP Poisson, %s, %d polytopes of %d^%d cells, %s(%d) n=%d, "
                                   "QGauss(%d), variant %s; %d dofs, %d nnz"
                                   % (args.dim,
                                      ("unit cube, %d^%d hex cells" % (args.cells, args.dim)) if (world == 1 or args.scaling == "strong")
                                      else ("%d unit cubes stacked along the last direction (one connected mesh), %s x %d hex cells"
                                            % (world, " x ".join([str(args.cells)] * (args.dim - 1)), args.cells * world)),
                                      r["n_agg"], args.block, args.dim,
                                      "FE_DGQ" if args.fe == "dgq" else "FE_AggloDGP", args.degree, r["n"],
                                      args.degree + 1, {"poisson": "examples/poisson.cc", "diffusion_reaction": "examples/diffusion_reaction.cc",
                                                        "assemble_dg_matrix": "PolyUtils::assemble_dg_matrix"}[args.variant],
                                      r["n_dofs"], r["nnz"]),
                       "algorithm": r["alg"], "rows_kernel": r["rows_kernel"],
                       "parallelism": ("rows(polytopes) split in %d contiguous ranges" % world if args.scaling == "strong" or world == 1
                                       else "rank r owns the rows of slab r (%d polytopes per rank)" % (r["n_agg"] // world))
                                      + ", rank-local descriptions, owner-computes-rows, no data-path collective",
                       "exchange_variant": r["ghost"], "per_rank": r["per_rank"]},
            "roofline": roof,
            "cpu_baseline": cpu,
            "extra": extra,
            "setup_s": r["t_setup"], "setup_parts": r["setup_parts"], "checksum": r["checksum"], "traffic_note": traffic_note,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
