#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -save-temps .s file (gfx950): per kernel, and per basic block holding
matrix instructions, the counts of MFMA / VALU / transcendental / SALU / LDS / VMEM / SMEM / waits / scratch traffic.
usage: isa_mix.py file.s [kernel-name-substring]"""
import collections
import re
import sys


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_accvgpr"):
        return "acc_copy"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("scratch_",)):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    name, block = None, None
    per_kernel = collections.OrderedDict()
    for line in open(path):
        line = line.rstrip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            per_kernel[name] = collections.OrderedDict()
            block = "entry"
            per_kernel[name][block] = collections.Counter()
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            block = m.group(1)
            per_kernel[name][block] = collections.Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if m and not line.strip().startswith((".", ";")):
            per_kernel[name][block][klass(m.group(1))] += 1
    for kname, blocks in per_kernel.items():
        if want not in kname:
            continue
        tot = collections.Counter()
        for c in blocks.values():
            tot.update(c)
        print(f"== {kname}\n   total: {dict(tot)}")
        for bname, c in blocks.items():
            if c.get("mfma", 0) >= 4:
                v = c.get("valu", 0) + c.get("trans", 0) + c.get("acc_copy", 0)
                print(f"   {bname}: {dict(c)}  -> VALU/MFMA = {v / c['mfma']:.2f}")


if __name__ == "__main__":
    main()
