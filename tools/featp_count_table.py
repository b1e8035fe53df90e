#!/usr/bin/env python3
"""k_app_features_p<false>: the instructions of ONE tile (32 appearance samples: every basic block of the tile loop in the compiler's
assembly, the exec-masked ones included) by mnemonic, priced with the per-class issue costs of bench.py's issue model (VALU 5, LDS 8,
MFMA 21 cycles; a 16-byte-per-lane gather 16 cycles of the texture addresser's 64 B/clk), and what of it is removable without changing
the arithmetic. Usage: featp_count_table.py t2n_shade.s"""
import collections
import re
import sys

WHY = {}


def klass(m):
    if m.startswith("v_mfma"):
        return "mfma", 21.0
    if m.startswith(("global_load", "buffer_load", "flat_load")):
        return "gather", 16.0 if "x4" in m else (8.0 if "x2" in m else 4.0)
    if m.startswith(("global_store", "buffer_store")):
        return "store", 16.0
    if m.startswith("ds_"):
        return "lds", 8.0
    if m.startswith("v_"):
        return "valu", 5.0
    if m.startswith("s_waitcnt"):
        return "wait", 1.0
    if m.startswith("s_"):
        return "salu", 1.0
    return "other", 1.0


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3t2n16k_app_features_pILb0EE"))
    end = next(i for i, l in enumerate(lines) if i > start and ".end_amdhsa_kernel" in l or (i > start and l.startswith(".Lfunc_end")))
    header = None
    for i in range(start, end):
        if "FEATP_MARK tile_begin" in lines[i]:
            j = i
            while not re.match(r"\.LBB\d+_\d+:", lines[j]):
                j -= 1
            header = re.match(r"\.(LBB\d+_\d+):", lines[j]).group(1)[1:]
            break
    in_loop, c = False, collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"\.(LBB\d+_\d+):(.*)", l)
        if m:
            in_loop = m.group(1)[1:] == header or f"Header={header}" in m.group(2)
            continue
        m2 = re.match(r"; %bb\.\d+:(.*)", l)
        if m2:
            in_loop = f"Header={header}" in m2.group(1)
            continue
        t = l.strip()
        if not in_loop or not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        c[t.split()[0]] += 1
    tot = collections.Counter()
    cyc = {}
    for k, v in c.items():
        kl, cost = klass(k)
        tot[kl] += v
        cyc[k] = v * cost
    total_cyc = sum(cyc.values())
    print("k_app_features_p<false>, one tile of one wave (32 samples x 144 channels): %d instructions, %.1f k issue cycles by the per-class "
          "cost table" % (sum(c.values()), total_cyc / 1e3))
    print("by class: " + ", ".join(f"{k} {v}" for k, v in tot.most_common()))
    print(f"{'mnemonic':30s}{'count':>7s}{'cycles each':>13s}{'k cycles':>10s}{'share':>8s}")
    for k, v in sorted(c.items(), key=lambda kv: -cyc[kv[0]]):
        if cyc[k] / total_cyc < 0.004:
            continue
        print(f"{k:30s}{v:7d}{klass(k)[1]:13.1f}{cyc[k] / 1e3:10.2f}{cyc[k] / total_cyc:8.1%}")


if __name__ == "__main__":
    main()
