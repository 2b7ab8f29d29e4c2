import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import polydeal_oracle as po
from flatten_oracle import flatten
import polydeal_amd as pa
from test_gpu_parity import build, gpu_values
fe = po.FE_AggloDGP(2, 7)
ah = build(2, 2, 2, fe, 8, distort=0.1)
var = po.variant_assemble_dg_matrix()
kw = flatten(ah, var)
_, _, ref = po.assemble_csr(ah, var)
got = gpu_values(kw)
n = fe.n_dofs_per_cell
A = po.csr_to_dense(kw["rowptr"], kw["colind"], np.where(np.isfinite(got), 0.0, 1.0), ah.n_dofs)
B = po.csr_to_dense(kw["rowptr"], kw["colind"], got, ah.n_dofs)
R = po.csr_to_dense(kw["rowptr"], kw["colind"], ref, ah.n_dofs)
o0, o1 = ah.dof_offset[0], ah.dof_offset[1]
blk = A[o0:o0+n, o1:o1+n]
for r in range(n):
    print("".join("X" if x else "." for x in blk[r]))
print("bad finite entries:", np.sum(np.isfinite(B) & (np.abs(B-R) > 1e-9*np.abs(R).max())))
