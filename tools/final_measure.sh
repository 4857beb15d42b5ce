#!/bin/bash
# End-of-round measurement on ONE GPU box (gpurun): tests, the bench line, per-kernel tables, rocprofv3 kernel stats, PMC traffic.
# Everything lands in gpurun_out/r05/; the judged copies are then committed under profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- "MSAU_GIT_SHA=$(git rev-parse --short HEAD) tools/final_measure.sh"
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05
mkdir -p "$O" && cd "$R" || exit 1
# (the GPU tests + smoke are their own gpurun call: tools/final_tests.sh -- together they no longer fit one 20-minute box)
# PMC traffic first (it stamps profiles/r05_traffic.json with this tree's kernel hash), then the bench line that reads it
python tools/make_traffic.py > "$O/traffic.log" 2>&1 || { tail -5 "$O/traffic.log"; exit 1; }
# SQ counters of every kernel inside the step (MFMA busy, VALU per MFMA): profiles/r05_pmc_step.txt, r05_pmc.json
python tools/pmc_step.py > "$O/pmc_step.log" 2>&1 || { tail -5 "$O/pmc_step.log"; exit 1; }
python bench.py --dump-kernels "$O/r05_final_hip_events.csv" > "$O/r05_final_bench.json" 2> "$O/bench.err" || { tail -5 "$O/bench.err"; exit 1; }
cut -c1-400 "$O/r05_final_bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o r05 -- python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 60 --warmup 10 > "$O/prof.log" 2>&1 || { tail -5 "$O/prof.log"; exit 1; }
MSAU_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_serial" -o r05s -- python3 "$R/bench.py" --no-secondary --no-cpu-baseline --no-roofline --steps 30 --warmup 10 > "$O/prof_serial.log" 2>&1 || { tail -5 "$O/prof_serial.log"; exit 1; }
bash "$R/tools/timeline.sh" "$O/tl" > /dev/null 2>&1; cp "$O/tl/timeline.txt" "$O/r05_timeline.txt" 2>/dev/null || true
find "$O/prof" "$O/prof_serial" -name "*kernel_trace.csv" -delete        # large; the stats and the timeline are what is kept
cp "$R/profiles/r05_traffic.json" "$R/profiles/r05_pmc.json" "$R/profiles/r05_pmc_step.txt" "$O/" 2>/dev/null
cd "$R"
ls "$O"
