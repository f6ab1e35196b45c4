"""Static checks of the generated gfx950 ISA, run by ``mneslam_amd.build`` on every kernel it compiles.

1. ``scan_asm`` -- the compiler defect behind DESIGN.md section 9 (ROCm 7.2 / clang-22 AMDGPU backend, found in round 3):
   the register allocator may place a VGPR *spill store* (``scratch_store_* ... Folded Spill``) at the top of a
   control-flow JOIN block, in front of the ``s_or_b64 exec, exec, s[..]`` that re-enables the lanes masked off by the
   preceding divergent region.  The store then only saves the lanes that were active inside the region (lane 0 after
   ``if (lane == 0) {...}``); the matching reload runs with every lane enabled and hands stale scratch contents to the
   others.  Wave-uniform values that live in VGPRs (a ray's decoded-sample count, ...) silently become garbage on 63
   lanes: results then depend on what earlier waves left in scratch -- non-deterministic, sensitive to anything that
   perturbs register allocation (the size of an unrelated kernel argument), invisible to the host emulator.
   A kernel without spill stores cannot be affected; a kernel with them is clean when no spill store precedes an
   exec-widening instruction inside its basic block.

2. ``resources`` -- registers / scratch / occupancy of every kernel, from the compiler's own "Kernel info" comments
   (the spill table under profiles/).
"""
import re

# widening for sure: s_or_b64 exec, exec, <saved mask>  (SI_END_CF) and s_or_saveexec (else entry re-enables the other half)
EXEC_WIDEN = re.compile(r"^\s*(s_or_b64 exec, exec,|s_or_saveexec_b64)")
# s_mov_b64 exec, sX is an if-entry (narrowing) when sX was just computed by s_and_b64 in this block, a restore otherwise
EXEC_MOV = re.compile(r"^\s*s_mov_b64 exec, (s\[\d+:\d+\])")
S_AND_DEF = re.compile(r"^\s*s_and(?:n2)?_b64 (s\[\d+:\d+\]),")
SPILL_STORE = re.compile(r"^\s*(scratch_store|buffer_store)\S*\s.*Folded Spill")
LABEL = re.compile(r"^(\.LBB\d+_\d+|[_A-Za-z][\w$.]*):")
TERMINATOR = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc)")


def scan_asm(text):
    """-> ({kernel: [(line_no, block, spill line, exec line, 'definite'|'possible')]}, {kernel: n_spill_stores})"""
    findings, spills = {}, {}
    kernel, block, pending, narrowed = None, None, [], set()
    for n, line in enumerate(text.split("\n"), 1):
        m = LABEL.match(line)
        if m:
            name = m.group(1)
            if not name.startswith(".LBB"):
                kernel = name if name.startswith("_Z") else kernel
            block, pending, narrowed = name, [], set()
            continue
        if kernel is None:
            continue
        d = S_AND_DEF.match(line)
        if d:
            narrowed.add(d.group(1))
        if SPILL_STORE.match(line):
            spills[kernel] = spills.get(kernel, 0) + 1
            pending.append((n, line.strip()))
            continue
        mv = EXEC_MOV.match(line)
        kind = None
        if EXEC_WIDEN.match(line):
            kind = "definite"
        elif mv and mv.group(1) not in narrowed:
            kind = "possible"
        if kind:
            for sn, sl in pending:
                findings.setdefault(kernel, []).append((sn, block, sl, line.strip(), kind))
            pending = []
        elif mv or TERMINATOR.match(line) or "s_and_saveexec" in line or "v_cmpx" in line or "s_and_b64 exec" in line:
            pending = []          # exec narrows from here on / the block ends: later stores are not in front of a restore
    return findings, spills


def definite_hazards(text):
    findings, _ = scan_asm(text)
    return [(k, f) for k, lst in findings.items() for f in lst if f[4] == "definite"]


_INFO = {"sgpr": r"; TotalNumSgprs: (\d+)", "vgpr": r"; NumVgprs: (\d+)", "agpr": r"; NumAgprs: (\d+)", "scratch": r"; ScratchSize: (\d+)",
         "occupancy": r"; Occupancy: (\d+)", "lds": r"; LDSByteSize: (\d+)", "sgpr_spill": r"\.sgpr_spill_count:\s+(\d+)",
         "vgpr_spill": r"\.vgpr_spill_count:\s+(\d+)"}


def resources(text):
    """-> [{kernel, sgpr, vgpr, agpr, scratch, occupancy, lds, sgpr_spill, vgpr_spill}] for every kernel in the file."""
    out = {}
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)", text, re.M):
        out[m.group(1)] = {"kernel": m.group(1)}
    # "Kernel info" comment blocks follow each function body
    for m in re.finditer(r"^(_Z\w+):.*?; Kernel info:(.*?)(?=^\s*\.(?:section|text|protected|globl|p2align))", text, re.M | re.S):
        name, info = m.group(1), m.group(2)
        if name in out:
            for key in ("sgpr", "vgpr", "agpr", "scratch", "occupancy", "lds"):
                mm = re.search(_INFO[key], info)
                out[name][key] = int(mm.group(1)) if mm else -1
    # spill counts live in the metadata (YAML) at the end of the file
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n\s+- \.|\namdhsa\.target|\Z)", text, re.S):
        name = m.group(1)
        if name in out:
            for key in ("sgpr_spill", "vgpr_spill"):
                mm = re.search(_INFO[key], m.group(0))
                if mm:
                    out[name][key] = int(mm.group(1))
    return list(out.values())
