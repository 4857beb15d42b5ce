#!/bin/bash
# one kernel-trace run of bench.py -> per-queue timeline of the last steps
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d $out/kt -- python3 bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $out/kt_bench.log 2>&1 || exit 1
db=$(find $out/kt -name "*.db" | head -1)
python3 tools/timeline.py $db --step-kernel adam --nsteps 20 --list > $out/timeline.txt 2>&1
rm -rf $out/kt
