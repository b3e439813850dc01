#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 600 python tools/r03_diag2.py > $O/diag2.log 2>&1; tail -60 $O/diag2.log
