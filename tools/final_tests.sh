#!/bin/bash
# The GPU test suite + smoke on one box, log to gpurun_out/r05/tests.log:  /usr/local/graft/bin/gpurun --timeout 1100 -- 'tools/final_tests.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05
mkdir -p "$O" && cd "$R" || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/tests.log"; tail -3 "$O/tests.log"
python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a "$O/tests.log"
