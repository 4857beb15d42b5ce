#!/bin/bash
# rocprofv3 kernel stats of bench.py under two (or more) environments on ONE box: tools/ab_stats.sh OUTDIR "ENV_A" "ENV_B" ...
# (each ENV is "K=V K=V" or "-"; the environment is exported in THIS shell: the profiled program stays directly behind `--`)
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  (
    if [ "$e" != "-" ]; then export $e; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p$i" -o s -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-secondary --no-cpu-baseline --no-roofline --steps 40 --warmup 10 > "$out/p$i.log" 2>&1
  ) || { tail -5 "$out/p$i.log"; exit 1; }
  f=$(find "$out/p$i" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$out/stats_$i.csv"
  find "$out/p$i" -name "*kernel_trace.csv" -delete
  echo "$i $e: $(tail -1 "$out/p$i.log" | cut -c1-120)"
done
