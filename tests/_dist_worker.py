"""Worker of tests/test_distributed_cpu.py: one rank of a gloo group (CPU).  Each rank builds the same
global problem with the product host mirror, takes its row range, runs the host-side validation/packing
of pdh_set_problem_local (pdh_check_problem: no GPU needed) and the ranks cross-check that their pieces
tile the global problem exactly."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polydeal_amd as pa  # noqa: E402
from polydeal_amd.partition import row_range  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    stacked = os.environ.get("PDH_DIST_STACK") == "1"
    if stacked:  # the mesh of bench.py's weak-scaling mode: one unit cube per rank, stacked along z, ONE connected problem
        grid = pa.BackgroundGrid.subdivided_hyper_rectangle(3, (4, 4, 4 * world), (0.0, 0.0, 0.0), (1.0, 1.0, float(world)))
    else:
        grid = pa.BackgroundGrid.hyper_cube_refined(3, 0.0, 1.0, 3)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(2)
    fe = pa.FE_AggloDGP(3, 2)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(fe)
    flat = ah.flatten(pa.SipVariant.diffusion_reaction(fe), True, True)
    n = fe.n_dofs_per_cell
    lib = pa.load_library()
    r0, r1 = row_range(ah.n_agglomerates, n, rank, world)
    st = (C.c_int64 * 8)()
    rc = lib.pdh_check_problem(C.byref(flat.c), r0, r1, st)
    assert rc == 0, lib.pdh_last_error(None)
    gl = (C.c_int64 * 8)()
    assert lib.pdh_check_problem(C.byref(flat.c), 0, flat.c.n_rows, gl) == 0
    mine = torch.tensor(list(st)[:5], dtype=torch.int64)
    tot = mine.clone()
    dist.all_reduce(tot)
    expect = torch.tensor(list(gl)[:5], dtype=torch.int64)
    # owned polytopes, volume points, own-side face points and values tile the global problem exactly;
    # coupling items: one per interior face, faces cut by the partition are computed on both ranks
    # (each writes only its own rows), so the sum exceeds the global count by the number of cut faces
    keep = [0, 2, 3, 4]
    assert torch.equal(tot[keep], expect[keep]), (tot, expect)
    assert int(tot[1]) > int(expect[1]) if world > 1 else int(tot[1]) == int(expect[1])
    ranges = [None] * world
    dist.all_gather_object(ranges, (r0, r1))
    assert ranges[0][0] == 0 and ranges[-1][1] == flat.c.n_rows
    for a, b in zip(ranges[:-1], ranges[1:]):
        assert a[1] == b[0]
    rp = flat.arrays()["rowptr"]
    assert int(st[4]) == int(rp[r1] - rp[r0])
    # a misaligned range must be rejected
    assert lib.pdh_check_problem(C.byref(flat.c), r0 + 1, r1, st) != 0
    # RANK-LOCAL description (owned + ghost polytopes only, global dof numbers, rowptr/colind of the owned rows): what an
    # MPI rank of the reference holds (source/agglomeration_handler.cc:1026-1091).  Packs to the same counts as the
    # global description restricted to the range, and its pattern is the slice of the global one.
    import numpy as np
    splits = [row_range(ah.n_agglomerates, n, r, world)[0] for r in range(world)] + [flat.c.n_rows]
    loc = ah.flatten_local(pa.SipVariant.diffusion_reaction(fe), r0, r1, True, True, row_splits=splits)
    assert loc.c.local == 1 and loc.c.n_rows == flat.c.n_rows and loc.c.n_agg < flat.c.n_agg or world == 1
    sl = (C.c_int64 * 8)()
    rc = lib.pdh_check_problem(C.byref(loc.c), r0, r1, sl)
    assert rc == 0, lib.pdh_last_error(None)
    assert list(sl) == list(st), (list(sl), list(st))
    ga, la = flat.arrays(), loc.arrays()
    assert np.array_equal(la["rowptr"], ga["rowptr"][r0:r1 + 1] - ga["rowptr"][r0])
    assert np.array_equal(la["colind"], ga["colind"][ga["rowptr"][r0]:ga["rowptr"][r1]])
    lof = loc.local_of()
    assert np.array_equal(la["dof_offset"], ga["dof_offset"][lof])
    assert np.array_equal(la["agg_rank"] == rank, (la["dof_offset"] >= r0) & (la["dof_offset"] < r1))
    if stacked:
        # slab r = rank r's rows: the local description names only the neighbouring slab(s) as ghosts, and qualifies for the
        # row kernel (host-only eligibility test: planar faces, tensor rules found on the points)
        assert ah.n_agglomerates == 8 * world and r1 - r0 == 8 * n
        n_ghost = int(loc.c.n_agg) - 8
        assert n_ghost == (4 if rank in (0, world - 1) else 8) or world == 1, n_ghost
        lib.pdh_check_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        assert lib.pdh_check_rows(C.byref(loc.c), r0, r1) == 1, lib.pdh_last_error(None)
    # ghost-block exchange variant: what this rank sends to a peer is what the peer expects from it (block order is derived
    # on both sides from the global dof numbers of the cut faces; only the sizes can be checked without a GPU)
    sc, rc_ = (C.c_int64 * world)(), (C.c_int64 * world)()
    assert lib.pdh_check_exchange(C.byref(loc.c), r0, r1, world, sc, rc_) == 0, lib.pdh_last_error(None)
    send = torch.tensor(list(sc), dtype=torch.int64)
    expect_recv = torch.zeros(world, dtype=torch.int64)
    dist.all_to_all_single(expect_recv, send)
    assert torch.equal(expect_recv, torch.tensor(list(rc_), dtype=torch.int64)), (expect_recv, list(rc_))
    tot_sent = send.sum().clone()
    dist.all_reduce(tot_sent)  # (with the id() < id() rule every cut face is owned by the lower rank: the last rank sends nothing)
    assert int(send[rank]) == 0 and (world == 1 or int(tot_sent) > 0)
    assert int(send.sum()) % (n * n) == 0
    # the local description must not be accepted for another rank's range
    if world > 1:
        o0, o1 = row_range(ah.n_agglomerates, n, (rank + 1) % world, world)
        assert lib.pdh_check_problem(C.byref(loc.c), o0, o1, sl) != 0
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
