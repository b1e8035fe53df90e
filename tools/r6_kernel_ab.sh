#!/bin/bash
# round 6: one kernel's average duration under the fused train loop, between library builds (rocprofv3 --kernel-trace --stats):
#   r6_kernel_ab.sh OUT KERNEL_SUBSTRING "LIBS ('' = shipped)" [BATCH]
out=gpurun_out/${1:-r6_kab}; pat=$2; libs=$3; batch=${4:-16384}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in main $libs; do
  if [ "$lib" = main ]; then unset T2N_LIB; else export T2N_LIB=$PWD/text2nerf_amd/libt2n_hip_$lib.so; fi
  rm -rf $out/prof_$lib
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$lib -o p -- python3 tools/experiments/train_only.py 2 40 $batch fused_eager > $out/log_$lib.txt 2>&1
  f=$(find $out/prof_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib" | tee -a $out/ab.txt
  python3 - "$f" "$pat" <<PY | tee -a $out/ab.txt
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("   calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1), "|", r["Name"][:70])
PY
  rm -rf $out/prof_$lib
done
