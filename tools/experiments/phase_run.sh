#!/bin/bash
# Per-phase wave cycles of the appearance kernels: a -DT2N_PHASE_TIMING build (text2nerf_amd/libt2n_hip_phase.so) selected through
# T2N_LIB; the shipped library stays untouched (on the GPU box).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
T2N_LIB=$PWD/text2nerf_amd/libt2n_hip_phase.so python tools/experiments/ss_phase.py
T2N_LIB=$PWD/text2nerf_amd/libt2n_hip_phase.so T2N_SHADE_NO_WS=1 python tools/experiments/phase_timing.py
