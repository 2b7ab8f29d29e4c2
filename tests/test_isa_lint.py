"""The generated gfx950 code around the library's inline-asm statements, audited on the build machine (tools/isa_lint.py): hipcc
neither counts the memory operations of an asm statement nor pads its hazards nor knows that it changes EXEC, and the GPU parity
suite only shows that the CURRENT register allocation happens to be safe.  Runs hipcc -S on the row / moment / term kernels
(cross-compilation, no GPU) and requires: no compiler instruction touching the destination of an asm load before its wait, every
asm statement that narrows EXEC saving and restoring the mask it found, no VALU-written SGPR consumed by an asm instruction inside
its wait states, and no scratch memory in the headline kernel."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-command-line-argument", "-S",
         "--cuda-device-only"]


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = {}
    d = tmp_path_factory.mktemp("isa")
    procs = []
    for src in ("pdh_moment.hip", "pdh_terms.hip"):
        dst = str(d / (src + ".s"))
        procs.append((src, dst, subprocess.Popen([HIPCC] + FLAGS + [os.path.join(ROOT, "polydeal_amd", "csrc", src), "-o", dst],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, dst, p in procs:
        log = p.communicate()[0].decode()
        assert p.returncode == 0, log[-2000:]
        out[src] = isa_lint.parse(dst)
    return out


def test_no_finding_in_any_row_moment_or_term_kernel(assembly):
    total, seen_exec_asm = [], 0
    for src, (kernels, meta) in assembly.items():
        assert kernels, src
        for name, body in kernels.items():
            f = isa_lint.lint_kernel(name, body)
            total += [(name, x) for x in f]
            seen_exec_asm += sum(1 for x in body if isinstance(x, isa_lint.Ins) and x.asm and x.text.startswith("s_mov_b64 exec, 1"))
    assert not total, total[:10]
    # since round 4 the carry reads of the shifted pieces are plain C++ (`if (lane == 0)`): NO asm statement of the library writes
    # EXEC any more; what is left are LDS reads with a deferred wait (general-point paths), waits and empty barriers
    assert seen_exec_asm == 0


def test_headline_kernel_runs_without_scratch_and_owns_no_exec_asm(assembly):
    kernels, meta = assembly["pdh_moment.hip"]
    name = "_ZN4pdhr6k_rowsILi4ELi0ELb0ELb1ELb0EEEv6PdhDev7PdhRowsPKdi"  # FE_DGQ(3), tensor rules, diagonal-first rows, one plane per neighbour
    assert name in kernels
    assert meta[name]["private_segment_fixed_size"] == 0 and meta[name]["vgpr_spill_count"] == 0, meta[name]
    body = [x for x in kernels[name] if isinstance(x, isa_lint.Ins)]
    assert not [x for x in body if x.asm and isa_lint.EXEC_WRITE.match(x.text)]
    # the compiler's own lane-0 regions around the carry reads: narrowed by s_and_saveexec, restored by s_or exec, reads paired
    assert sum(1 for x in body if x.op == "ds_read2_b64") >= 24 * 4
    term, tmeta = assembly["pdh_terms.hip"]
    for k, m in tmeta.items():  # the term kernel has no asm at all; its 4-point instantiations run without scratch
        if "ELi4EEEv" in k:
            assert m["private_segment_fixed_size"] == 0, (k, m)
    assert not [x for body_ in term.values() for x in body_ if isinstance(x, isa_lint.Ins) and x.asm]


def test_lint_flags_what_it_is_meant_to_flag(tmp_path):
    """The three rules on hand-made snippets (a linter that never fires proves nothing)."""
    src = tmp_path / "t.s"
    src.write_text("""
	.type	k,@function
k:
	s_and_saveexec_b64 s[0:1], vcc
	;;#ASMSTART
	s_mov_b64 exec, 1
	ds_read_b64 v[2:3], v1
	s_mov_b64 exec, -1
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	;;#ASMSTART
	ds_read_b64 v[4:5], v1 offset:8
	;;#ASMEND
	v_mov_b32_e32 v9, v4
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_readlane_b32 s7, v255, 3
	;;#ASMSTART
	global_store_dwordx2 v8, v[2:3], s[6:7]
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
""")
    kernels, _ = isa_lint.parse(str(src))
    kinds = sorted(k for k, _, _ in isa_lint.lint_kernel("k", kernels["k"]))
    assert kinds == ["A", "B", "C"], kinds
