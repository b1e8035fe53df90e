#!/usr/bin/env python3
"""Instruction census of k_mlp_ss3 per region (the `; SS3_MARK` comments the kernel source plants): MFMA, other VALU, LDS, VMEM, SALU,
s_nop (states), s_waitcnt, s_barrier — from the compiler's .s (hipcc -S --cuda-device-only). Usage: ss3_isa_count.py file.s"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3t2n2ss9k_mlp_ss3"))
end = next(i for i, l in enumerate(lines) if i > start and ".end_amdhsa_kernel" in l)
region, counts, order = "prologue", collections.defaultdict(collections.Counter), ["prologue"]
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; SS3_MARK (\w+)", t)
    if m:
        region = m.group(1)
        if region not in order:
            order.append(region)
        continue
    if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
        continue
    op = t.split()[0]
    c = counts[region]
    if op.startswith("v_mfma"):
        c["mfma"] += 1
    elif op == "s_nop":
        c["s_nop"] += 1
        c["nop_states"] += int(t.split()[1]) + 1
    elif op == "s_waitcnt":
        c["s_waitcnt"] += 1
    elif op == "s_barrier":
        c["s_barrier"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        c["vmem"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
        if op.startswith("v_accvgpr"):
            c["accvgpr"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    else:
        c["other"] += 1
keys = ["mfma", "valu", "accvgpr", "lds", "vmem", "salu", "s_nop", "nop_states", "s_waitcnt", "s_barrier", "other"]
print(f"{'region':10s}" + "".join(f"{k:>11s}" for k in keys) + f"{'issue':>9s}{'per mfma':>10s}")
for r in order:
    c = counts[r]
    issue = c["mfma"] + c["valu"] + c["lds"] + c["vmem"] + c["salu"] + c["s_nop"] + c["s_waitcnt"] + c["s_barrier"]
    print(f"{r:10s}" + "".join(f"{c[k]:11d}" for k in keys) + f"{issue:9d}{(issue / c['mfma'] if c['mfma'] else 0):10.2f}")
