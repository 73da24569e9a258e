#!/bin/bash
# full GPU suite + smoke after the link / colsum / frag_pair / rank-128 bench changes
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s27
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s27/pytest.log 2>&1 && tail -3 gpurun_out/s27/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')"
