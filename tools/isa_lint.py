#!/usr/bin/env python3
"""Static audit of the gfx950 assembly hipcc generates AROUND the library's inline-asm statements.

hipcc treats an `asm volatile` statement as one opaque instruction: it neither counts the memory operations inside nor pads
their hazards, and it does not know that a statement changes EXEC (guide section 5.7).  The parity suite only shows that the
CURRENT register allocation is safe; this tool checks the generated code itself, so that a change of compiler, flags or
register pressure that breaks one of the assumptions below is seen on the build machine (tests/test_isa_lint.py runs it on the
shipped sources; no GPU needed):

  A  deferred waits.  An asm LDS / global load whose `s_waitcnt` is a LATER asm statement (pdh_moment.h: lds_read + vwait /
     fwait; pdh_rows.h P2): between the load and its wait the compiler believes the destination VGPRs are already written.
     Any compiler instruction in that window that reads one of them (copy, spill, use: stale data) or writes one of them (the
     load lands afterwards and overwrites the new value - a pointer or a spilled SGPR lane, in the worst case) is an error.
  B  EXEC.  An asm statement that writes EXEC (pdh_rows.h: lane 0 alone reads the carries) must restore what it found: either
     it saves / restores the mask itself, or - if it restores the constant -1 - it must not sit in a region where the
     compiler's own code has narrowed EXEC.
  C  wait states at the boundary.  A VALU instruction that writes an SGPR (v_readlane_b32 restoring a spilled SGPR,
     v_readfirstlane_b32, v_cmp into an SGPR pair) within 4 / 5 issue slots in front of an asm instruction that reads that
     SGPR as lane select of v_readlane / v_writelane (4 wait states) or as address / descriptor / offset of a memory
     instruction (5).  (Both instructions compiler-generated: the hazard recogniser pads them; it cannot pad an asm string.)
  D  scratch.  Kernels that are expected to run without scratch memory (--no-scratch REGEX) must have none.

usage: isa_lint.py FILE.s [--kernels REGEX] [--no-scratch REGEX] [--quiet]      exit status 1 if anything is flagged."""
import argparse
import re
import sys

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]")
MEM_LOAD = re.compile(r"^(ds_read|ds_load|global_load|buffer_load|scratch_load|flat_load)")
MEM_ANY = re.compile(r"^(ds_|global_|buffer_|scratch_|flat_)")
WAITCNT = re.compile(r"^s_waitcnt\b")
EXEC_WRITE = re.compile(r"^s_\w+_saveexec_b64\b|^(s_\w+)\s+exec(_lo|_hi)?\b|^v_cmpx")
SGPR_WRITERS = re.compile(r"^(v_readlane_b32|v_readfirstlane_b32|v_cmp_\w+|v_add_co_u32|v_addc_co_u32|v_sub_co_u32|v_subb_co_u32|v_mad_u64_u32|v_div_scale_f64)\b")
BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+|s_endpgm|s_setpc_b64|s_swappc_b64)\b")


def regs(text, kind):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            if m.group(1) == kind:
                out.add(int(m.group(2)))
        elif m.group(3) == kind:
            out.update(range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_operands(line):
    parts = line.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])]
    return parts[0], ops


class Ins:
    __slots__ = ("text", "op", "ops", "asm", "lineno", "stmt")

    def __init__(self, text, asm, lineno, stmt):
        self.text, self.asm, self.lineno, self.stmt = text, asm, lineno, stmt
        self.op, self.ops = split_operands(text)

    def dest_text(self):
        return self.ops[0] if self.ops else ""

    def src_text(self):
        return ", ".join(self.ops[1:]) if len(self.ops) > 1 else ""


def parse(path):
    """-> {kernel name: (list of Ins / ('label', name), {metadata})}"""
    kernels, cur, name, in_asm, stmt = {}, None, None, False, 0
    meta = {}
    functions = set()
    for n, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.strip()
        if not line:
            continue
        mt = re.match(r"^\.type\s+(\w+),@function", line)
        if mt:
            functions.add(mt.group(1))
            continue
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and cur is None and raw[0] not in " \t" and m.group(1) in functions:
            name, cur = m.group(1), []
            continue
        if cur is None:
            # (kernel metadata, YAML at the end of the file: the fields of a kernel follow its .name)
            m3 = re.match(r"^\.name:\s*(\S+)", line)
            if m3:
                meta["_cur"] = m3.group(1)
                meta.setdefault(m3.group(1), {})
            m2 = re.match(r"^\.(private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count|vgpr_count|sgpr_count):\s*(\d+)", line)
            if m2 and "_cur" in meta:
                meta[meta["_cur"]][m2.group(1)] = int(m2.group(2))
            continue
        if line.startswith(".Lfunc_end"):
            kernels[name] = cur
            cur = None
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm, stmt = True, stmt + 1
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if line.startswith(";") or line.startswith("//"):
            continue
        if re.match(r"^\.?L?\w+:$", line) or re.match(r"^\.LBB\w+:", line):
            cur.append(("label", line.rstrip(":")))
            continue
        if line.startswith("."):
            continue
        text = line.split(";")[0].strip()
        if text:
            cur.append(Ins(text, in_asm, n, stmt if in_asm else 0))
    meta.pop("_cur", None)
    return kernels, meta


def lint_kernel(name, body):
    findings = []
    ins = [x for x in body]
    # ---------- A: asm loads whose wait is a later asm statement
    i = 0
    n = len(ins)
    while i < n:
        x = ins[i]
        if isinstance(x, Ins) and x.asm and MEM_LOAD.match(x.op):
            # the statement: all consecutive asm instructions with the same statement number
            j = i
            pend = set()
            waited = False
            while j < n and isinstance(ins[j], Ins) and ins[j].asm and ins[j].stmt == x.stmt:
                if MEM_LOAD.match(ins[j].op):
                    pend |= regs(ins[j].dest_text(), "v")
                if WAITCNT.match(ins[j].op):
                    waited = True
                    pend.clear()
                j += 1
            if pend and not waited:
                first_line = x.lineno
                k = j
                while k < n:
                    y = ins[k]
                    if isinstance(y, tuple):
                        findings.append(("A", first_line, "asm load of v%s still pending at the label %s (control flow leaves the window)" % (sorted(pend), y[1])))
                        break
                    if y.asm:
                        if WAITCNT.match(y.op) and ("lgkmcnt(0)" in y.text or "vmcnt(0)" in y.text):
                            break
                        if MEM_LOAD.match(y.op):
                            pend |= regs(y.dest_text(), "v")
                        k += 1
                        continue
                    if BRANCH.match(y.op):
                        findings.append(("A", first_line, "asm load of v%s still pending at `%s` (line %d)" % (sorted(pend), y.text, y.lineno)))
                        break
                    if WAITCNT.match(y.op) and "lgkmcnt(0)" in y.text:
                        break  # (a compiler wait for its own LDS traffic drains ours as well)
                    touched = regs(y.text, "v") & pend
                    if touched:
                        findings.append(("A", y.lineno, "`%s` touches v%s between an asm load (line %d) and its wait" % (y.text, sorted(touched), first_line)))
                    k += 1
            i = j
            continue
        i += 1
    # ---------- B: asm statements that write EXEC
    for i, x in enumerate(ins):
        if not (isinstance(x, Ins) and x.asm and EXEC_WRITE.match(x.text)):
            continue
        if i > 0 and isinstance(ins[i - 1], Ins) and ins[i - 1].asm and ins[i - 1].stmt == x.stmt and any(
                isinstance(z, Ins) and z.asm and z.stmt == x.stmt and EXEC_WRITE.match(z.text) for z in ins[max(0, i - 40):i]):
            continue  # not the first EXEC write of its statement
        stmt = [z for z in ins[i:i + 64] if isinstance(z, Ins) and z.asm and z.stmt == x.stmt]
        writes = [z for z in stmt if EXEC_WRITE.match(z.text)]
        saves = [z for z in ins[max(0, i - 4):i + 1] if isinstance(z, Ins) and z.asm and z.stmt == x.stmt and re.match(r"^s_(mov|and_saveexec|or_saveexec)_b64\s+s\[\d+:\d+\],\s*exec", z.text)]
        saves += [z for z in stmt if re.match(r"^s_(and|or)_saveexec_b64\s+s\[\d+:\d+\]", z.text) or re.match(r"^s_mov_b64\s+s\[\d+:\d+\],\s*exec", z.text)]
        last = writes[-1]
        restores_saved = bool(re.match(r"^s_mov_b64\s+exec,\s*s\[\d+:\d+\]", last.text))
        if restores_saved and saves:
            continue
        if not re.match(r"^s_mov_b64\s+exec,\s*-1", last.text):
            findings.append(("B", x.lineno, "asm statement leaves EXEC as `%s`" % last.text))
            continue
        # restores the constant -1: the compiler's own code in front must not have narrowed EXEC
        k = i - 1
        narrowed = None
        while k >= 0:
            y = ins[k]
            if isinstance(y, Ins) and not y.asm and EXEC_WRITE.match(y.text):
                if re.match(r"^s_or_b64\s+exec,\s*exec", y.text) or re.match(r"^s_mov_b64\s+exec,\s*-1", y.text):
                    break  # end of a divergent region: full mask again
                narrowed = y
                break
            k -= 1
        if narrowed is not None:
            findings.append(("B", x.lineno, "asm sets EXEC = -1 after the compiler's `%s` (line %d): lanes that were off are switched on" % (narrowed.text, narrowed.lineno)))
    # ---------- C: VALU-written SGPR -> asm consumer within the wait states
    for i, x in enumerate(ins):
        if not (isinstance(x, Ins) and x.asm):
            continue
        need, sg = 0, set()
        if re.match(r"^v_(readlane|writelane)_b32", x.op) and len(x.ops) >= 3:
            need, sg = 4, regs(x.ops[2], "s")
        elif MEM_ANY.match(x.op):
            need, sg = 5, regs(x.text, "s")
        if not sg:
            continue
        slots, k = 0, i - 1
        while k >= 0 and slots < need:
            y = ins[k]
            if isinstance(y, tuple):
                break
            if y.op == "s_nop":
                try:
                    slots += int(y.ops[0], 0) + 1
                except (ValueError, IndexError):
                    slots += 1
                k -= 1
                continue
            if not y.asm and SGPR_WRITERS.match(y.op):
                hit = set()
                for d in y.ops[:2] if y.op.startswith(("v_add_co", "v_sub_co", "v_addc", "v_subb", "v_mad_u64", "v_div_scale")) else y.ops[:1]:
                    hit |= regs(d, "s")
                if hit & sg:
                    findings.append(("C", x.lineno, "`%s` (line %d) writes s%s %d slot(s) in front of the asm `%s`, which needs %d wait states"
                                     % (y.text, y.lineno, sorted(hit & sg), slots, x.text, need)))
            slots += 1
            k -= 1
    return findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernels", default=".", help="regex on the (mangled) kernel name")
    ap.add_argument("--no-scratch", default=None, help="regex: these kernels must not use scratch memory")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    kernels, meta = parse(a.asm)
    bad = 0
    sel = re.compile(a.kernels)
    for name in sorted(kernels):
        if not sel.search(name):
            continue
        body = kernels[name]
        n_asm = len({x.stmt for x in body if isinstance(x, Ins) and x.asm})
        f = lint_kernel(name, body)
        m = meta.get(name + ".kd", meta.get(name, {}))
        if a.no_scratch and re.search(a.no_scratch, name) and m.get("private_segment_fixed_size", 0):
            f.append(("D", 0, "uses %d bytes of scratch per lane" % m["private_segment_fixed_size"]))
        if f or not a.quiet:
            print("%s: %d instructions, %d asm statements, %d finding(s)%s" % (name, sum(isinstance(x, Ins) for x in body), n_asm, len(f),
                  (" [vgpr %s sgpr spills %s scratch %s]" % (m.get("vgpr_count"), m.get("sgpr_spill_count"), m.get("private_segment_fixed_size"))) if m else ""))
        for kind, line, msg in f:
            print("  %s line %d: %s" % (kind, line, msg))
        bad += len(f)
    print("%d finding(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
