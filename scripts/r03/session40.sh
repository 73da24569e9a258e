#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in libbmf_hip.so libbmf_stag25.so libbmf_stag50.so; do echo "== $lib"; BMF_LIB=$lib timeout -k 10 200 python scripts/r03/link_wide_bench.py pnlpf kl 2>&1 | grep "it/s"; done
