"""C ABI checks that need no GPU: the library loads, exports every symbol include/polydeal_hip.h
declares, validates problem descriptions, and refuses to compute without a HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import polydeal_amd as pa
from polydeal_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "polydeal_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pdh_[a-z_]+)\s*\(", hdr))
    declared -= {"pdh_ctx", "pdh_problem"}
    assert len(declared) >= 17
    lib = pa.load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert set(_capi.EXPORTS) == declared
    assert b"gfx950" in lib.pdh_version()


def small_problem(dim=2, basis="dgp", p=2, refine=2):
    grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, refine)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(2)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(dim, p)
    ah.initialize_fe_values(p + 1, p + 1)
    ah.distribute_agglomerated_dofs(fe)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, True)
    kw = {k: (None if v is None else np.array(v)) for k, v in flat.arrays().items()}
    c = flat.c
    kw.update(dim=c.dim, degree=c.degree, basis=c.basis, n_agg=c.n_agg, n_faces=c.n_faces, n_rows=c.n_rows,
              diag_first=c.diag_first, reaction_c=c.reaction_c)
    return kw


def test_check_problem_accepts_valid_and_reports_stats():
    kw = small_problem()
    st = pa.Problem(**kw).check()
    assert st[0] == kw["n_agg"] and st[4] == kw["rowptr"][-1] and st[5] == 6
    assert st[1] == int(np.sum(kw["face_out"] >= 0))  # one coupling item per interior face (A[Q,P] = A[P,Q]^T)
    assert st[2] == kw["vq_ptr"][-1]


@pytest.mark.parametrize("mutate,code", [
    (lambda k: k.update(dim=4), _capi.PDH_EINVAL),
    (lambda k: k.update(basis=7), _capi.PDH_EINVAL),
    (lambda k: k.update(n_rows=k["n_rows"] + 1), _capi.PDH_EINVAL),
    (lambda k: k.__setitem__("dof_offset", k["dof_offset"] + 1), _capi.PDH_EINVAL),
    (lambda k: k.__setitem__("bbox", np.zeros_like(k["bbox"])), _capi.PDH_EINVAL),
    (lambda k: k.__setitem__("face_out", np.where(k["face_out"] >= 0, k["n_agg"] + 3, -1)), _capi.PDH_EINVAL),
    (lambda k: k.__setitem__("rowptr", k["rowptr"] * 2), _capi.PDH_EINVAL),
    (lambda k: k.__setitem__("colind", k["colind"][::-1].copy()), _capi.PDH_EINVAL),
    (lambda k: k.__setitem__("vq_w", None), _capi.PDH_EINVAL),
])
def test_check_problem_rejects_malformed(mutate, code):
    kw = small_problem()
    mutate(kw)
    with pytest.raises(pa.PdhError) as ei:
        pa.Problem(**kw).check()
    assert ei.value.code == code


def test_unsupported_block_size_is_reported_not_silently_wrong():
    kw = small_problem(dim=3, basis="dgq", p=4, refine=1)  # n = 125 > 64
    with pytest.raises(pa.PdhError) as ei:
        pa.Problem(**kw).check()
    assert ei.value.code == _capi.PDH_EUNSUPPORTED


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pa.PdhError) as ei:
        pa.Context(0)
    assert ei.value.code == _capi.PDH_EDEVICE
    grid = pa.BackgroundGrid.hyper_cube_refined(2, 0.0, 1.0, 1)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(1)
    fe = pa.FE_DGQ(2, 1)
    ah.initialize_fe_values(2, 2)
    ah.distribute_agglomerated_dofs(fe)
    with pytest.raises(pa.HostError):
        pa.assemble_dg_matrix(fe, ah)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polydeal_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "sip_ref" not in src, f
