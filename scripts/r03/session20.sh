#!/bin/bash
# the whole GPU suite + smoke() on the current build
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s20; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/test.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
