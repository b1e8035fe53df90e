#!/bin/bash
# Per-phase wave cycles of k_shade_coop: swaps a -DT2N_PHASE_TIMING build in as libt2n_hip.so for one run (on the GPU box).
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp text2nerf_amd/libt2n_hip.so /tmp/libt2n_hip.so.orig
cp text2nerf_amd/libt2n_hip_phase.so text2nerf_amd/libt2n_hip.so
python tools/experiments/phase_timing.py || true
cp /tmp/libt2n_hip.so.orig text2nerf_amd/libt2n_hip.so
