#!/bin/bash
# A/B of library builds on ONE box: each arm = bench.py --no-cpu-baseline with LMX_LIB pointing at a build.
# usage: bash tools/ab_bench.sh out_name "label=lib.so[:ENV=VAL...]" ...
set -u
OUT=gpurun_out/$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for rep in 1 2; do
for arm in "$@"; do
  label=${arm%%=*}; rest=${arm#*=}
  lib=${rest%%:*}; envs=""
  if [[ "$rest" == *:* ]]; then envs=$(echo "${rest#*:}" | tr ':' ' '); fi
  line=$(env $envs LMX_LIB=$PWD/vision-sam3-yolo-lameless_amd/lmx/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1)
  echo "$label rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "gemm TF", round(d["roofline"]["achieved"],1), "serial", d["roofline"]["measured_on"][-28:])')" | tee -a $OUT
done
done
