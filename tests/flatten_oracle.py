"""Test helper: flatten an ORACLE handler into the arrays of a pdh_problem (so the HIP path and the
oracle see bit-identical inputs).  Test infrastructure, not product code."""
import numpy as np

from oracle import polydeal_oracle as po


def flatten(ah, var, diag_first=True, with_colind=True):
    dim = ah.grid.dim
    fe = ah.fe
    nA = ah.n_agglomerates
    bbox = np.zeros((nA, 2, dim))
    vq_ptr = [0]
    xs, ws = [], []
    for P in range(nA):
        bbox[P, 0], bbox[P, 1] = ah.bboxes[P]
        x, w = ah.agglomerated_quadrature(P)
        xs.append(x)
        ws.append(w)
        vq_ptr.append(vq_ptr[-1] + len(w))
    vq_x = np.concatenate(xs).T.copy()  # [dim][Nq]
    vq_w = np.concatenate(ws)
    face_in, face_out, fq_ptr, sig = [], [], [0], []
    fx, fn, fw, fwo = [], [], [], []
    for P in range(nA):
        for f in range(ah.n_faces[P]):
            if ah.at_boundary(P, f):
                if var.boundary == "zero":
                    continue
                ff = ah.reinit_face(P, f)
                face_in.append(P)
                face_out.append(-1)
                sig.append(po.face_sigma(ah, var, P))
                fx.append(ff["x"]); fn.append(ff["normal"]); fw.append(ff["JxW"]); fwo.append(ff["JxW"])
            else:
                Q = ah.neighbor(P, f)
                if po._owns(ah, var, P, Q):
                    nofn = ah.neighbor_of_agglomerated_neighbor(P, f)
                    f0, f1 = ah.reinit_interface(P, Q, f, nofn)
                    face_in.append(P)
                    face_out.append(Q)
                    sig.append(po.face_sigma(ah, var, P, Q))
                    fx.append(f0["x"]); fn.append(f0["normal"]); fw.append(f0["JxW"]); fwo.append(f1["JxW"])
                else:
                    continue
            fq_ptr.append(fq_ptr[-1] + len(fw[-1]))
    nF = len(face_in)
    rowptr, colind = ah.sparsity_pattern(diag_first)
    kw = dict(
        dim=dim, degree=fe.degree, basis=fe.basis_id, n_agg=nA, n_faces=nF, n_rows=ah.n_dofs,
        diag_first=int(diag_first), reaction_c=var.reaction_c,
        bbox=bbox, dof_offset=ah.dof_offset, vq_ptr=vq_ptr, vq_x=vq_x, vq_w=vq_w,
        rowptr=rowptr, colind=colind if with_colind else None,
    )
    if nF:
        kw.update(face_in=face_in, face_out=face_out, fq_ptr=fq_ptr,
                  fq_x=np.concatenate(fx).T.copy(), fq_n=np.concatenate(fn).T.copy(),
                  fq_w=np.concatenate(fw), fq_w_out=np.concatenate(fwo), face_sigma=sig)
    return kw
