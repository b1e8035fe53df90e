#!/bin/bash
# Run a script against an experimental build of the library: swap_run.sh <lib.so> <script.py> [args]. The build is selected with T2N_LIB
# (text2nerf_amd/_lib.py); the shipped libt2n_hip.so stays untouched.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
lib=$(realpath "$1"); shift
T2N_LIB=$lib python "$@"
