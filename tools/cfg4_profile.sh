#!/bin/bash
# cfg 4 (BASELINE configs[3]: 336x256x768 BERT-embedding chargrid, 2 stages): bench line, per-kernel HIP-event table and
# rocprofv3 kernel stats.  Output in gpurun_out/r02cfg4/; judged copies go to profiles/r02_cfg4_*.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'tools/cfg4_profile.sh'
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02cfg4
mkdir -p "$O" && cd "$R" || exit 1
A="--channels 768 --stages 2 --no-secondary --no-cpu-baseline"
python bench.py $A --steps 60 --warmup 10 --dump-kernels "$O/r02_cfg4_hip_events.csv" > "$O/r02_cfg4_bench.json" 2> "$O/bench.err" || { tail -5 "$O/bench.err"; exit 1; }
cut -c1-600 "$O/r02_cfg4_bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o cfg4 -- python3 "$R/bench.py" $A --no-roofline --steps 40 --warmup 10 > "$O/prof.log" 2>&1 || { tail -5 "$O/prof.log"; exit 1; }
find "$O/prof" -name "*kernel_trace.csv" -delete
ls -R "$O" | head
