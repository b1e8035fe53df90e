#!/usr/bin/env python3
"""Timing-only variants of k_mlp_ss3 made by EDITING THE COMPILER'S ASSEMBLY between the kernel's `; SS3_MARK` region comments, to tell
which instruction class the single wave's stalls come from (results are garbage: run with bench.py --quick, look at the time only).

    ss3_asm_variants.py build NAME=EDIT[@REGIONS] ...     -> text2nerf_amd/libt2n_hip_NAME.so (base: -DSS3_PROF -DSS3_TIMING_ONLY)
EDITs (comma separated): nonop (drop s_nop), nolgkm (drop s_waitcnt lgkmcnt(N > 0)), noaccread (v_accvgpr_read -> v_mov of a VGPR),
dropacc (drop v_accvgpr_read), nolds (drop ds_read), novalu (drop every VALU that is not an MFMA / accvgpr), nomfma (drop MFMAs), novmem (drop buffer/global ops),
base (no edit). REGIONS: any of layer0,stage0,stage1,stage2,tail (default stage0..tail)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "text2nerf_amd", "csrc", "t2n_mlp_ss.hip")
OUT = os.path.join(ROOT, "build", "ss3asm")
LLVM = "/opt/rocm/lib/llvm/bin"
CF = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics", "-DNDEBUG",
      "-DSS3_PROF", "-DSS3_TIMING_ONLY"]


def sh(*cmd):
    subprocess.run(list(cmd), check=True)


def edit(lines, edits, regions):
    out, region, inside = [], "prologue", False
    for l in lines:
        t = l.strip()
        if t.startswith("_ZN3t2n2ss9k_mlp_ss3ENS0_4ArgsE:"):
            inside = True
        if inside and ".end_amdhsa_kernel" in t:
            inside = False
        m = re.match(r"; SS3_MARK (\w+)", t)
        if m:
            region = m.group(1)
        if not inside or region not in regions or not t or t.startswith((";", ".")) or t.endswith(":"):
            out.append(l)
            continue
        op = t.split()[0]
        if "nonop" in edits and op == "s_nop":
            continue
        if "nolgkm" in edits and op == "s_waitcnt" and "vmcnt" not in t and "lgkmcnt" in t and "lgkmcnt(0)" not in t:
            continue   # (lgkmcnt(0) stays: it may guard a scalar load of a pointer)
        if op.startswith(("global_store", "buffer_store", "global_atomic")) and edits != {"base"}:
            continue   # garbage values / addresses are never written
        if "nolds" in edits and op.startswith("ds_read"):
            continue
        if "novmem" in edits and op.startswith(("buffer_", "global_")):
            continue
        if "nomfma" in edits and op.startswith("v_mfma"):
            continue
        if "dropacc" in edits and op == "v_accvgpr_read_b32":
            continue
        if "noaccread" in edits and op == "v_accvgpr_read_b32":
            dst = t.split()[1].rstrip(",")
            out.append(f"\tv_mov_b32_e32 {dst}, {dst}")
            continue
        if "novalu" in edits and op.startswith("v_") and not op.startswith(("v_mfma", "v_accvgpr", "v_readfirstlane", "v_cmp", "v_cndmask")):
            continue
        out.append(l)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    base_s = os.path.join(OUT, "base.s")
    sh("/opt/rocm/bin/hipcc", *CF, "-S", "--cuda-device-only", SRC, "-o", base_s)
    lines = open(base_s).read().split("\n")
    others = [os.path.join(ROOT, "build", "obj", o) for o in sorted(os.listdir(os.path.join(ROOT, "build", "obj"))) if o.endswith(".o") and o != "t2n_mlp_ss.o"]
    for spec in sys.argv[2:]:
        name, _, rest = spec.partition("=")
        ed, _, reg = rest.partition("@")
        edits = set(ed.split(",")) if ed else {"base"}
        regions = set(reg.split("+")) if reg else {"stage0", "stage1", "stage2", "tail"}
        s = os.path.join(OUT, name + ".s")
        open(s, "w").write("\n".join(edit(lines, edits, regions)))
        o, hsaco, fb, ho = (os.path.join(OUT, name + e) for e in (".dev.o", ".hsaco", ".hipfb", ".host.o"))
        sh(f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o)
        sh(f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, o)
        sh(f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
           "-input=/dev/null", f"-input={hsaco}", f"-output={fb}")
        sh("/opt/rocm/bin/hipcc", *CF, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", SRC, "-o", ho)
        lib = os.path.join(ROOT, "text2nerf_amd", f"libt2n_hip_{name}.so")
        sh("/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", *others, ho, "-o", lib)
        print("built", lib, "edits", sorted(edits), "regions", sorted(regions), flush=True)


if __name__ == "__main__":
    main()
