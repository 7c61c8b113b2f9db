#!/bin/bash
# runs tools/stamps.py with each diagnostic library given (tools/diag_build.sh): sweeps-phase cycles per variant
for n in "$@"; do
  echo "=== HS_DIAG=$n"
  HSFLOW_LIB_PATH=tools/bin/libhsflow_diag$n.so python tools/stamps.py --configs "20:5:1024;12:4:1024" 2>&1 | grep -E "^T=|sweeps|total"
done
