#!/bin/bash
# Decomposition of attn_gp_kernel (csrc/attn.hip, -DLMX_GP_DBG=n development builds: make O=obj_gpN EXTRA=-DLMX_GP_DBG=N LIB=../lmx/liblmx_gpN.so):
# the time of the two large shapes with one part of the loop removed (results are wrong by construction).
cd "$(dirname "$0")/.." || exit 1
L=vision-sam3-yolo-lameless_amd/lmx
echo "# full kernel"; python tools/attn_gp_probe.py child | head -2
for v in "1 no v_exp_f32 (the argument is passed through)" "2 no S MFMAs / K fragment reads" "3 no PV MFMAs / V fragment reads" "4 no row maximum" "5 no LDS-DMA in the loop"; do
  n=${v%% *}
  [ -f $L/liblmx_gp$n.so ] || continue
  echo "# build ${v}"
  LMX_LIB=$PWD/$L/liblmx_gp$n.so python tools/attn_gp_probe.py child
done
