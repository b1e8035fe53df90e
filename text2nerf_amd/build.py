"""Build libt2n_hip.so (gfx950) in-tree with hipcc. `python -m text2nerf_amd.build [--force]`.

Every csrc/*.hip is compiled to its own object (in parallel, cached under build/obj by source + header mtimes) and the objects
are linked into the shared library: a one-file edit rebuilds in seconds, a cold build in ~15 s instead of a minute."""
from __future__ import annotations

import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libt2n_hip.so")
OBJ = os.path.join(ROOT, "build", "obj")

CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-DNDEBUG"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wl,--no-undefined"]   # an unresolved internal symbol must fail the build, not the first dlopen on the GPU box
FLAGS = CFLAGS + LDFLAGS[1:]      # (kept for callers that print the flag set)
# per-file additions. t2n_mlp_bwd_ss.hip: the SLP vectoriser packs the (sin, cos) chains of neighbouring features into v_pk_* pairs,
# which makes it evaluate every chain up front and spill them (604 B of scratch per lane against 76 B without)
PER_FILE = {"t2n_mlp_bwd_ss.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "t2n.h"), __file__]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return os.path.getmtime(src) > t or any(os.path.getmtime(h) > t for h in _headers())


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def check_ss3_isa(hipcc, verbose=False, extra=(), obj_dir=None):
    """k_mlp_ss3 keeps live accumulators in literally named AccVGPRs across separate asm statements; nothing reserves them in
    between, so the build is refused should the compiler ever put values of its own there or spill: the kernel's assembly must hold no
    v_accvgpr_write / v_accvgpr_mov (hipcc staging a value in an AccVGPR), no scratch, no spilled registers. One more -S pass of that
    file. ADVICE r5: the verdict is remembered by a stamp file next to the object (`t2n_mlp_ss.checked`, newer than the object = passed);
    build() re-runs the check whenever the stamp is missing or older, and removes a refused object so that no later build links it."""
    import re
    obj_dir = obj_dir or OBJ
    src = os.path.join(CSRC, "t2n_mlp_ss.hip")
    asm = os.path.join(obj_dir, "t2n_mlp_ss.s")
    subprocess.run([hipcc] + CFLAGS + list(extra) + ["-S", "--cuda-device-only", src, "-o", asm], check=True,
                   stderr=subprocess.DEVNULL)
    text = open(asm).read()
    found = 0
    for m in re.finditer(r"^(_ZN3t2n2ss9k_mlp_ss3\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        found += 1
        bad = [l.strip() for l in body.split("\n") if re.match(r"\s*v_accvgpr_(write|mov)", l)]
        meta = re.search(r"\.name:\s+" + name + r"\n.*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
        priv = re.search(r"\.name:\s+" + name + r"\n.*?\.private_segment_fixed_size:\s+(\d+)", text, re.S)
        spills = int(meta.group(1)) if meta else -1
        scratch = int(priv.group(1)) if priv else -1
        if bad or spills != 0 or scratch != 0:
            raise RuntimeError(f"{name}: the compiler touched the AccVGPRs or spilled ({len(bad)} v_accvgpr_write/mov, {spills} spilled "
                               f"registers, {scratch} B of scratch): the hand-allocated accumulators are not safe with this build")
        if verbose:
            print(f"check_ss3_isa: {name} ok ({body.count('v_accvgpr_read_b32')} accumulator reads, no writes, no spills)", flush=True)
    if found != 2:
        raise RuntimeError(f"check_ss3_isa: expected the two instantiations of k_mlp_ss3 in the assembly, found {found}")


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in sources() if force or _stale(s)]

    def compile_one(src):
        cmd = [hipcc] + CFLAGS + PER_FILE.get(os.path.basename(src), []) + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    ss_obj = _obj(os.path.join(CSRC, "t2n_mlp_ss.hip"))
    stamp = ss_obj[:-2] + ".checked"
    if not os.path.exists(stamp) or os.path.getmtime(stamp) < os.path.getmtime(ss_obj):
        try:
            check_ss3_isa(hipcc, verbose)
        except Exception:
            for f in (ss_obj, stamp):         # a refused object must not survive into the next build()'s link
                if os.path.exists(f):
                    os.remove(f)
            raise
        open(stamp, "w").close()
    keep = {_obj(s) for s in sources()}
    for o in glob.glob(os.path.join(OBJ, "*.o")):       # objects of sources that no longer exist must not be linked
        if o not in keep:
            os.remove(o)
    cmd = [hipcc] + LDFLAGS + sorted(keep) + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    if "--check-ss3" in sys.argv:          # tools/build_variant.sh: python -m text2nerf_amd.build --check-ss3 OBJ_DIR [extra flags]
        i = sys.argv.index("--check-ss3")
        check_ss3_isa(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), True, extra=sys.argv[i + 2:], obj_dir=sys.argv[i + 1])
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
