"""Static check of the generated gfx950 ISA for the compiler defect behind DESIGN.md section 9.

    python profiles/spill_exec_scan.py [file.hip ...] [-- extra hipcc flags]     (default: every kernel source)

Defect (ROCm 7.2 / clang-22 AMDGPU backend, found in round 3): the register allocator may place a VGPR *spill store*
(`scratch_store_* ... Folded Spill`) at the top of a control-flow JOIN block, in front of the `s_or_b64 exec, exec, s[..]`
that re-enables the lanes masked off by the preceding divergent region.  The store then only saves the lanes that were
active inside the region (e.g. lane 0 after `if (lane == 0) {...}`); the matching reload runs with every lane enabled and
hands stale scratch contents to the others.  Wave-uniform values that live in VGPRs (a ray's decoded-sample count, ...)
silently become garbage on 63 lanes -- results then depend on what earlier waves left in scratch: non-deterministic,
layout-sensitive, invisible to the host emulator.

The scan walks every kernel: inside each basic block, any spill store that precedes an `s_or_b64 exec, exec` /
`s_mov_b64 exec` / `s_or_saveexec` of the same block (block prologue) is reported.  Kernels without spill stores cannot
be affected.  Exit status 1 if anything is found.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "mneslam_amd", "csrc")
INC = os.path.join(HERE, "..", "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]

# widening for sure: s_or_b64 exec, exec, <saved mask>  (SI_END_CF) and s_or_saveexec (else entry re-enables the other half)
EXEC_WIDEN = re.compile(r"^\s*(s_or_b64 exec, exec,|s_or_saveexec_b64)")
# s_mov_b64 exec, sX is an if-entry (narrowing) when sX was just computed by s_and_b64 in this block, a restore otherwise
EXEC_MOV = re.compile(r"^\s*s_mov_b64 exec, (s\[\d+:\d+\])")
S_AND_DEF = re.compile(r"^\s*s_and(?:n2)?_b64 (s\[\d+:\d+\]),")
SPILL_STORE = re.compile(r"^\s*(scratch_store|buffer_store)\S*\s.*Folded Spill")
LABEL = re.compile(r"^(\.LBB\d+_\d+|[_A-Za-z][\w$.]*):")
TERMINATOR = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc)")


def scan_asm(text):
    """-> ({kernel: [(line_no, block, spill line, restore line, 'definite'|'possible')]}, {kernel: n_spill_stores})"""
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
            pending = []          # exec narrows from here on / block ends: later stores are not in front of a restore
    return findings, spills


def asm_of(src, extra):
    out = "/tmp/_scan_%s.s" % os.path.basename(src)
    subprocess.check_call(["hipcc", *FLAGS, *extra, "-I", INC, "-I", CSRC, "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    asm_files = [a for a in args if a.endswith(".s")]
    srcs = [a for a in args if not a.endswith(".s")] or ([] if asm_files else
            [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")])
    bad = 0
    for path in asm_files + srcs:
        text = open(path).read() if path.endswith(".s") else asm_of(path, extra)
        findings, spills = scan_asm(text)
        n_def = sum(1 for lst in findings.values() for f in lst if f[4] == "definite")
        print("== %s: %d kernels with VGPR spill stores; spill stores in front of an exec restore: %d definite, %d possible"
              % (os.path.basename(path), len(spills), n_def, sum(len(v) for v in findings.values()) - n_def))
        for k in sorted(spills):
            lst = findings.get(k, [])
            print("   %-90s spill stores %4d  definite %d possible %d" % (k[:90], spills[k], sum(1 for f in lst if f[4] == "definite"),
                                                                          sum(1 for f in lst if f[4] == "possible")))
        for k, lst in findings.items():
            bad += sum(1 for f in lst if f[4] == "definite")
            for sn, blk, sl, rl, kind in lst[:8]:
                print("   %s %s  line %d in %s:  %s   <- before ->   %s" % ("!!" if kind == "definite" else " ?", k[:60], sn, blk, sl, rl))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
