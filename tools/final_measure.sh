#!/bin/bash
# End-of-round measurement on ONE GPU box (gpurun): tests, the bench line, per-kernel tables, rocprofv3 kernel stats, PMC traffic.
# Everything lands in gpurun_out/r03/; the judged copies are then committed under profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'tools/final_measure.sh'
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03
mkdir -p "$O" && cd "$R" || exit 1
python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/tests.log"; tail -3 "$O/tests.log"
python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a "$O/tests.log"
# PMC traffic first (it stamps profiles/r03_traffic.json with this tree's kernel hash), then the bench line that reads it
python tools/make_traffic.py > "$O/traffic.log" 2>&1 || { tail -5 "$O/traffic.log"; exit 1; }
python bench.py --dump-kernels "$O/r03_final_hip_events.csv" > "$O/r03_final_bench.json" 2> "$O/bench.err" || { tail -5 "$O/bench.err"; exit 1; }
cut -c1-400 "$O/r03_final_bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o r03 -- python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 60 --warmup 10 > "$O/prof.log" 2>&1 || { tail -5 "$O/prof.log"; exit 1; }
MSAU_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_serial" -o r03s -- python3 "$R/bench.py" --no-secondary --no-cpu-baseline --no-roofline --steps 30 --warmup 10 > "$O/prof_serial.log" 2>&1 || { tail -5 "$O/prof_serial.log"; exit 1; }
find "$O/prof" "$O/prof_serial" -name "*kernel_trace.csv" -delete        # large; the stats are what is kept
cd "$R"
ls "$O"
