#!/usr/bin/env python3
"""Build step (orienmask_amd/csrc/Makefile) and test helper: scan the gfx950 code hipcc emitted for one source file
(build/NAME-hip-amdgcn-amd-amdhsa-gfx950.s, kept by -save-temps=obj) for the instruction form behind the erratum found in round 5
(tools/hazard_probe/pk_opsel_repro.hip, profiles/r05_experiments.md section 2): v_pk_add/mul/fma_f32 with a source-half selection
(op_sel / op_sel_hi) on a REGISTER operand returns wrong lanes now and then while another wave of the SIMD issues wide-K matrix
instructions.  Constants may be half-selected, registers not.  Exit status 1 (the build fails) when a file contains one.

    python3 isa_audit.py build/*.s
"""
import re
import sys

_SRC = re.compile(r"(v\[\d+:\d+\]|s\[\d+:\d+\]|-?\d+\.?\d*|0x[0-9a-f]+|v\d+|s\d+)")


def bad_instructions(path):
    bad = []
    for line in open(path):
        ins = line.split(";")[0].strip()
        if not re.match(r"v_pk_(add|mul|fma)_f32", ins) or "op_sel" not in ins:
            continue
        n = 3 if "fma" in ins else 2
        sel = re.search(r"op_sel:\[([0-9,]+)\]", ins)
        selhi = re.search(r"op_sel_hi:\[([0-9,]+)\]", ins)
        s = [int(v) for v in sel.group(1).split(",")] if sel else [0] * n
        sh = [int(v) for v in selhi.group(1).split(",")] if selhi else [1] * n
        srcs = _SRC.findall(ins.split(None, 1)[1])[1:1 + n]
        if any(i < len(srcs) and srcs[i].startswith("v[") and (s[i] != 0 or sh[i] != 1) for i in range(n)):
            bad.append(ins)
    return bad


def main(paths):
    failed = False
    for p in paths:
        bad = bad_instructions(p)
        if bad:
            failed = True
            sys.stderr.write("isa_audit: %s: %d packed fp32 instruction(s) with a register half-select, e.g. %s\n" % (p, len(bad), bad[0]))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
