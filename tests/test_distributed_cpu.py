"""N>1 logic on CPU: world_size-2 gloo group, row partition + per-rank packing (no GPU needed)."""
import os
import subprocess
import sys

from polydeal_amd.partition import polytope_range, row_range


def test_partition_is_a_tiling():
    for n_agg in (1, 7, 64, 32768):
        for world in (1, 2, 3, 8):
            if world > n_agg:
                continue
            prev = 0
            for r in range(world):
                a0, a1 = polytope_range(n_agg, r, world)
                assert a0 == prev and a1 >= a0
                prev = a1
                assert row_range(n_agg, 20, r, world) == (a0 * 20, a1 * 20)
            assert prev == n_agg


def test_two_rank_gloo():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
         "127.0.0.1", "--master-port", "29531", os.path.join(root, "tests", "_dist_worker.py")],
        env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIST_OK world=2" in out.stdout


def test_two_rank_gloo_stacked_mesh():
    """The mesh of bench.py's weak-scaling mode (N unit cubes stacked along z, slab r = rank r) through the same worker."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", PDH_DIST_STACK="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
         "127.0.0.1", "--master-port", "29533", os.path.join(root, "tests", "_dist_worker.py")],
        env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIST_OK world=2" in out.stdout
