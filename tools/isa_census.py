#!/usr/bin/env python3
"""Static instruction census of a gfx950 kernel from hipcc's assembly (-S --cuda-device-only): the kernel's outermost loop (largest
back-edge span) or the whole kernel, by instruction class and mnemonic. Used for the per-kernel count tables under profiles/ (round 5:
k_march_tiles, k_mlp_ss3).   tools/isa_census.py file.s kernel-name-substring [--whole] [--top N] [--range a b]"""
import collections
import re
import sys


def kernel_lines(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and name in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
    return [l.strip() for l in lines[start + 1:end + 1]]


def klass(m):
    if "mfma" in m:
        return "mfma"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if m.startswith("s_"):
        return "salu" if not m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")) else m.split("_")[1] if m.startswith(("s_nop", "s_waitcnt", "s_barrier")) else "branch"
    if m in ("v_exp_f32_e32", "v_log_f32_e32", "v_rcp_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32", "v_sin_f32_e32", "v_cos_f32_e32"):
        return "valu-trans"
    if m.startswith(("v_cvt_", "v_accvgpr", "v_mov_b32", "v_fma_mixlo", "v_fma_mixhi")) or m in ("v_mul_lo_u32", "v_mul_hi_u32") or m.startswith(("v_mad_u64", "v_mad_i64")):
        return "valu-slow"       # ~8 issue cycles next to an MFMA (profiles/round4_issue_cost_microbench.txt) / quarter-rate integer
    return "valu"


def main():
    path, name = sys.argv[1], sys.argv[2]
    whole = "--whole" in sys.argv
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    L = kernel_lines(path, name)
    labels = {l.split(":")[0]: i for i, l in enumerate(L) if re.match(r"^\.LBB\d+_\d+:", l)}
    a, b = 0, len(L)
    if "--range" in sys.argv:
        k = sys.argv.index("--range")
        a, b = int(sys.argv[k + 1]), int(sys.argv[k + 2])
    elif not whole:
        best = (0, 0, 0)
        for i, l in enumerate(L):
            if l.startswith(("s_cbranch", "s_branch")):
                t = l.split()[-1]
                if t in labels and labels[t] < i and i - labels[t] > best[0]:
                    best = (i - labels[t], labels[t], i + 1)
        a, b = best[1], best[2]
    if "--blocks" in sys.argv:      # basic blocks of the range: size, classes, closing branch
        blocks, cur, nm = [], [], "entry"
        for l in L[a:b]:
            if re.match(r"^\.LBB\d+_\d+:", l):
                if cur:
                    blocks.append((nm, cur))
                nm, cur = l.split(":")[0], []
                continue
            if not l or l.startswith((".", ";", "//")):
                continue
            cur.append(l)
            if l.startswith(("s_cbranch", "s_branch")):
                blocks.append((nm, cur))
                nm, cur = nm + "'", []
        if cur:
            blocks.append((nm, cur))
        for nm, ins in blocks:
            c = collections.Counter(klass(x.split()[0]) for x in ins)
            br = ins[-1] if ins and ins[-1].startswith(("s_cbranch", "s_branch")) else ""
            print(f"{nm:14s} {len(ins):4d}  valu {c['valu'] + c['valu-slow'] + c['valu-trans']:4d} salu {c['salu']:3d} vmem {c['vmem']:2d} "
                  f"lds {c['lds']:2d} mfma {c['mfma']:2d}  {br}")
    ins = [l for l in L[a:b] if l and not l.startswith((".", ";", "//")) and not re.match(r"^\S+:", l)]
    by_class, by_m = collections.Counter(), collections.Counter()
    for l in ins:
        m = l.split()[0]
        by_class[klass(m)] += 1
        by_m[m] += 1
    print(f"{name}: lines [{a}, {b}) of {len(L)}: {len(ins)} instructions")
    print("  by class:", ", ".join(f"{k} {v}" for k, v in by_class.most_common()))
    print("  top mnemonics:", ", ".join(f"{k} {v}" for k, v in by_m.most_common(top)))


if __name__ == "__main__":
    main()
