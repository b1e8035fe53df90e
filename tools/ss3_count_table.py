#!/usr/bin/env python3
"""k_mlp_ss3: instructions of one round (12 tiles per workgroup, 3 per wave: everything between the SS3_MARK layer0 and end markers of
the compiler's assembly) by mnemonic, priced with the issue costs measured next to an MFMA with one wave per SIMD
(profiles/round4_issue_cost_microbench.txt, profiles/round5_mixlo_cost_microbench.txt), and what of it could be removed without
changing the arithmetic. Usage: ss3_count_table.py t2n_mlp_ss.s"""
import collections
import re
import sys

COST = {  # issue cycles per instruction beside an MFMA, one wave per SIMD
    "v_mfma_f32_32x32x16_f16": 21.0, "v_cvt_pkrtz_f16_f32": 7.9, "v_fma_mix_f32": 4.6, "ds_read_b128": 10.0, "v_fma_f32": 4.5,
    "s_nop": 2.0, "v_max_f32_e32": 4.1, "s_waitcnt": 1.0, "v_accvgpr_read_b32": 7.9, "v_mul_f32_e32": 4.5, "v_add_f32_e32": 4.5,
    "v_pk_max_i16": 5.0, "v_fmac_f32_e32": 4.5, "v_fract_f32_e32": 4.1, "v_sin_f32_e32": 8.1, "v_cos_f32_e32": 8.1, "buffer_load_dwordx4": 8.0,
}
WHY = {
    "v_mfma_f32_32x32x16_f16": "3 f16 products per fp32 product x (351 x 128 + 128 x 128 + 128 x 32) / (32 x 32 x 16) per 32-sample tile: the arithmetic",
    "v_cvt_pkrtz_f16_f32": "2 per pair of values (hi halves, lo halves): the f16 operand format. v_fma_mixlo/mixhi_f16 build the lo pair in 2 "
                           "instead of 3 instructions but cost 8.6 cycles each (measured this round): 25 cycles per pair either way",
    "v_fma_mix_f32": "2 per pair: the residuals x - hi",
    "ds_read_b128": "A operands: 2 per element (hi, lo); layer 0 shares one read over 9 MFMAs, layers 1-2 over 3",
    "v_fma_f32": "scale + bias of the accumulators (1 per value), encoder polynomial steps",
    "v_max_f32_e32": "ReLU, 1 per hidden value",
    "v_accvgpr_read_b32": "1 per value of the accumulators kept in AccVGPRs (layer 0 of the three tiles, layer 1 of tile 1): the "
                          "architectural half of the file is full (238 registers)",
    "v_pk_max_i16": "REMOVABLE: f16-range tracking of the hidden activations, 1 per pair; a static bound from the weights' row norms would "
                    "replace it (second instantiation of the kernel + device-side selection)",
    "s_nop": "hazard states the assembler places between VALU-written operands and the MFMAs (dropping all of them in a timing-only "
             "build: -1.5 %)",
    "v_sin_f32_e32": "octaves 0 and 3 of the 27 features (the other four by double-angle steps)", "v_cos_f32_e32": "same",
}


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3t2n2ss9k_mlp_ss3"))
    end = next(i for i, l in enumerate(lines) if i > start and ".end_amdhsa_kernel" in l)
    region, c = "prologue", collections.Counter()
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"; SS3_MARK (\w+)", t)
        if m:
            region = m.group(1)
            continue
        if region in ("prologue", "end") or not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
            continue
        c[t.split()[0]] += 1
    total_cyc = sum(v * COST.get(k, 4.5) for k, v in c.items())
    print("k_mlp_ss3, one round of one wave (3 tiles = 96 samples): %d instructions, %.1f k issue cycles by the additive cost table "
          "(measured: 51.2 k cycles per round, matrix pipe alone 36.9 k)" % (sum(c.values()), total_cyc / 1e3))
    print(f"{'mnemonic':28s}{'count':>7s}{'cycles each':>13s}{'k cycles':>10s}{'share':>8s}  what it is")
    for k, v in c.most_common():
        cyc = v * COST.get(k, 4.5)
        if cyc / total_cyc < 0.004:
            continue
        print(f"{k:28s}{v:7d}{COST.get(k, 4.5):13.1f}{cyc / 1e3:10.2f}{cyc / total_cyc:8.1%}  {WHY.get(k, '')}")
    rem = c.get("v_pk_max_i16", 0) * COST["v_pk_max_i16"]
    print("removable without touching the arithmetic: v_pk_max_i16 %.2f k cycles = %.1f %% of the round's issue sum" % (rem / 1e3, 100 * rem / total_cyc))


if __name__ == "__main__":
    main()
