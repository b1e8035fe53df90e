#!/bin/bash
# r2_train_variants.sh NAME...: kernel time of the fused train step per iteration (rocprofv3 --stats) with
# text2nerf_amd/libt2n_hip_NAME.so selected through T2N_LIB; FILTER = substring of the kernels to list
root=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  lib=$root/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$root/text2nerf_amd/libt2n_hip.so
  out=$root/gpurun_out/tv_$v; rm -rf $out; mkdir -p $out
  cd /tmp && export TMPDIR=/tmp && cd $root
  T2N_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/experiments/train_only.py 2 40 > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" "${FILTER:-gemm}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = 43.0
tot = sum(float(r["TotalDurationNs"]) for r in rows if "k_march<false" not in r["Name"] and "k_mlp_ss" not in r["Name"] and "k_app_features" not in r["Name"]) / it / 1e3
sel = {r["Name"].split("(")[0][-28:]: round(float(r["TotalDurationNs"]) / it / 1e3, 1) for r in rows if sys.argv[3] in r["Name"]}
print(sys.argv[2], "sum of kernels per iteration: %.1f us" % tot, sel)
PY
done
