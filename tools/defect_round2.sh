#!/bin/bash
# Round-2 experiments on the wrong-value defect.  Run on the GPU box:  bash tools/defect_round2.sh
# Needs `make -C vision-sam3-yolo-lameless_amd/csrc dbg` and tools/pk_hazard_probe built in the container.
set -u
OUT=gpurun_out/defect
mkdir -p $OUT
L=vision-sam3-yolo-lameless_amd/lmx
T="timeout -k 10 240"
$T ./tools/pk_hazard_probe 1024 20000 > $OUT/pk_hazard.txt 2>&1 || echo "pk probe rc $?" >> $OUT/pk_hazard.txt
cat $OUT/pk_hazard.txt
: > $OUT/coresidency.txt
for spec in "dbg 1 attn" "dbg_noslp 1 attn" "dbg 7 attn" "dbg_noslp 7 attn" "dbg 1 none" "dbg_noslp 1 attn_128"; do
  set -- $spec
  LMX_LIB=$PWD/$L/liblmx_$1.so LMX_DBG_MASK=$2 FG=maskp $T python tools/coresidency_probe.py $3 2>&1 | tail -1 | sed "s/^/lib=$1 /" >> $OUT/coresidency.txt || exit 1
done
cat $OUT/coresidency.txt
