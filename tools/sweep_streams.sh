#!/bin/bash
# bench.py (dense schedule only) over SAM pass sizes x stream counts, one box, interleaved twice
OUT=gpurun_out/${1:-sweep_streams.txt}; : > $OUT
for rep in 1 2; do
for cfg in "16 6" "16 4" "8 6" "8 10" "10 8" "25 8" "32 6" "16 8"; do
  set -- $cfg
  v=$(LMX_MAX_STREAMS=$2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-reference-schedule --steps 4 --warmup 1 --sam-chunk $1 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))')
  echo "sam_chunk=$1 streams=$2 rep$rep: $v" | tee -a $OUT
done
done
