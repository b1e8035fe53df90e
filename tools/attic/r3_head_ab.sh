#!/bin/bash
# head-kernel A/B on one box: correctness of the main build on the head's parity tests, then per-kernel ms for each variant (twice, interleaved)
out=gpurun_out/${1:-r3_head_ab}; shift
mkdir -p $out
( python -m pytest tests/test_hip_parity.py tests/test_hip_range.py tests/test_hip_fullsize.py tests/test_list_budget.py tests/test_kept_rows_marker.py -x -q -m gpu -k "g5 or range or whole_frame or big300 or budget or kept or g6 or g7" ) > $out/tests.log 2>&1
tail -8 $out/tests.log
for rep in 1 2; do bash tools/r2_variants.sh "$@" 2>&1 | tee -a $out/variants.txt; done
[ -x tools/experiments/mfma_two_tile.bin ] && timeout 200 ./tools/experiments/mfma_two_tile.bin > $out/mfma_two_tile.txt 2>&1 && grep "tiles/wave 3" $out/mfma_two_tile.txt
