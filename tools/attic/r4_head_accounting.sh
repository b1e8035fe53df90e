#!/bin/bash
# Issue-slot accounting of the head kernel k_mlp_ss3: run on the GPU box after
#   bash tools/r4_ss3_variants.sh build prof=-DSS3_PROF "noenc=-DSS3_PROF -DSS3_ABL_NO_ENC" "noaf=-DSS3_PROF -DSS3_ABL_NO_AFETCH" \
#        "nodma=-DSS3_PROF -DSS3_ABL_NO_DMA" "nobar=-DSS3_PROF -DSS3_ABL_NO_BARRIER" "noconv=-DSS3_PROF -DSS3_ABL_NO_CONV" \
#        "s0a=-DSS3_PROF -DSS3_ABL_S0_NOCONV -DSS3_ABL_NO_ENC"
# (timing-only builds of t2n_mlp_ss.hip: their pictures are wrong, their range flags are suppressed). Writes gpurun_out/<dir>/accounting.txt:
# per-wave cycle sums of the kernel's regions (s_memtime) for the shipped code and every ablation, the instruction census of the
# shipped code per region (tools/ss3_isa_count.py over the compiler's .s) and the two-wave kernel on the same box.
out=gpurun_out/${1:-r4_head_accounting}; mkdir -p $out
{
  echo "# k_mlp_ss3: cycle sums per wave and region (s_memtime; 1024 waves, 43.6 rounds of 12 tiles per wave; C2 frame, scene S1-soft)"
  for v in prof noenc noaf nodma nobar noconv s0a; do
    lib=$PWD/text2nerf_amd/libt2n_hip_$v.so
    [ -f $lib ] || continue
    T2N_LIB=$lib python bench.py --no-train --steps 40 --quick --no-cpu-baseline 2>$out/err_$v.txt >/dev/null
    echo "$v: $(grep -h 'ss3 prof' $out/err_$v.txt | tail -1)"
  done
  echo
  echo "# same box, shipped library: kernel ms per frame (bench.py --quick), three-tile kernel, then T2N_SS_TWO_WAVE=1 where the build still has it"
  for rep in 1 2 3; do
    python bench.py --no-train --steps 100 --quick --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('k_mlp_ss3', 'ms/step', round(d['ms_per_step'], 3), {k: round(x, 4) for k, x in d['config']['kernel_ms_per_frame'].items()})"
  done
  echo
  echo "# instruction census of the shipped kernel per region (compiler output, tools/ss3_isa_count.py)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -DNDEBUG -Iinclude -S --cuda-device-only text2nerf_amd/csrc/t2n_mlp_ss.hip -o $out/k_mlp_ss.s 2>/dev/null
  python3 tools/ss3_isa_count.py $out/k_mlp_ss.s
} 2>&1 | tee $out/accounting.txt
rm -f $out/k_mlp_ss.s
