#!/bin/bash
# A/B of library variants on one box: parity subset on the main build, then per-kernel ms for each variant (twice, interleaved)
out=gpurun_out/${1:-r3_ab}; shift
mkdir -p $out
( python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_fuzz.py tests/test_list_budget.py -x -q -m gpu -k "tile_marcher or whole_frame or random_frames or budget or c4_shape or alpha_mask" ) > $out/tests.log 2>&1
tail -4 $out/tests.log
for rep in 1 2 3; do bash tools/r2_variants.sh "$@" 2>&1 | tee -a $out/variants.txt; done
