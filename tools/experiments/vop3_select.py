#!/usr/bin/env python3
"""vop3_select.py — rewrite `v_cndmask_b32_e32 vD, src0, vS, vcc` (VOP2: the mask is the implicit VCC) as `v_cndmask_b32_e64 vD, src0, vS, vcc`
(VOP3: VCC named as an ordinary SGPR pair) in the device assembly hipcc emits for gfx950. Same operation, four more bytes.

Why (tools/ubench/valu_rate.hip, measured on MI355X, profiles/r05_valu_rate.txt): a VOP2 v_cndmask whose VCC was not written by the instruction
just before it issues in 16-23 cycles instead of 4 — one compare followed by fifteen selects on it: 16.2 cycles per instruction in the VOP2
encoding, 4.06 in VOP3, 4.09 with the mask in s[10:11]. The compiler shrinks every select whose mask it could place in VCC to VOP2
(si-shrink-instructions; no switch turns that off), and a reservoir merge is a dozen selects on ONE comparison.

usage: vop3_select.py file.s   (rewritten in place; prints how many instructions changed)"""
import re
import sys

PAT = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+v\d+\s*,\s*[^,]+,\s*v\d+\s*,\s*vcc\s*)$")

def rewrite(path):
    changed = 0
    out = []
    for line in open(path):
        body = line.rstrip("\n")
        code, sep, comment = body.partition(";")
        m = PAT.match(code.rstrip())
        if m:
            body = f"{m.group(1)}v_cndmask_b32_e64{m.group(2)}" + (" " + sep + comment if sep else "")
            changed += 1
        out.append(body + "\n")
    open(path, "w").writelines(out)
    return changed

if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(f"{p}: {rewrite(p)} selects re-encoded as VOP3")
