#!/bin/bash
# Interleaved A/B of bench.py on ONE box: tools/ab_bench.sh OUTDIR ROUNDS "ENV_A" "ENV_B" ...   (each ENV is "K=V K=V" or "-")
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    if [ "$e" = "-" ]; then e=""; fi
    env $e python bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 150 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$i', '$e', d['ms_per_step'], d['value'])" | tee -a "$out/ab.log" || exit 1
  done
done
