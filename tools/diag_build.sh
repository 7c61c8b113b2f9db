#!/bin/bash
# Diagnostic builds of the library for timing experiments (wrong results!): tools/bin/libhsflow_diag<N>.so with
# -DHS_DIAG=<N> (bit mask, see hs_kernels_strip.hip.h) and only the R = 4 / 5 / 6 strip kernels (fast to compile).
# Use with HSFLOW_LIB_PATH=tools/bin/libhsflow_diag<N>.so python tools/stamps.py ...
set -e
cd "$(dirname "$0")/../opticalflowhs_amd/csrc"
mkdir -p ../../tools/bin
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine \
      -Wno-unused-value -DHS_DIAG_MIN -DHS_SWEEP_STAMPS=1 -DHS_DIAG=$n -o ../../tools/bin/libhsflow_diag$n.so hsflow.hip pair_pipeline.cpp multi_gpu.cpp &
done
wait
ls -la ../../tools/bin/
