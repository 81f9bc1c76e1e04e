#!/bin/bash
# Headline configuration through the drop-in CLI (reference README.md:82 command shape), synthetic NIF weights.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
A=/tmp/assets.extra; mkdir -p $A
python3 - <<PY
import sys; sys.path.insert(0, "$ROOT")
from ipu_path_trace_amd import nif_assets as N
N.write_metadata("$A/nif_metadata.txt"); N.write_ptnif("$A/converted.ptnif", N.synthetic_nif(), 12)
PY
make -C $ROOT/ipu_path_trace_amd/host -s
$ROOT/ipu_path_trace_amd/host/ipu_trace --assets $A -w 1104 -h 1000 -s 1500 --samples-per-step 300 --ipus 1 --defer-attach \
   -o /tmp/image.png --save-interval 10 --save-exe pt_graph --max-path-length 8 ${CLI_EXTRA}
