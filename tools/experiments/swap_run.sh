#!/bin/bash
# Run a script with an experimental build swapped in as libt2n_hip.so (on the GPU box): swap_run.sh <lib.so> <script.py> [args]
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp text2nerf_amd/libt2n_hip.so /tmp/libt2n_hip.so.orig
cp "$1" text2nerf_amd/libt2n_hip.so
shift
python "$@" || true
cp /tmp/libt2n_hip.so.orig text2nerf_amd/libt2n_hip.so
