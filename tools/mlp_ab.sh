# A/B of the fused LN + MLP variants on one box (csrc/mlp.hip): LMX_MLP_RING = 0 (product), 1 (fragment-read ring), 2 (ring, D = 112 at two workgroups per CU)
for r in 0 1 2 0 1; do echo "ring $r"; LMX_MLP_RING=$r timeout -k 10 200 python tools/mlp_probe.py 2>&1 | grep rows; done
