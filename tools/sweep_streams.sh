#!/bin/bash
# bench.py (dense schedule only) over SAM pass sizes x stream counts, one box, interleaved twice
# usage: bash tools/sweep_streams.sh out.txt "chunk streams" ...
OUT=gpurun_out/${1:-sweep_streams.txt}; shift; : > $OUT
for rep in 1 2; do
for cfg in "$@"; do
  c=${cfg% *}; s=${cfg#* }
  v=$(LMX_MAX_STREAMS=$s timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-reference-schedule --steps 4 --warmup 1 --sam-chunk $c 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))')
  echo "sam_chunk=$c streams=$s rep$rep: $v" | tee -a $OUT
done
done
