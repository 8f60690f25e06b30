# A/B of the fused LN + MLP variants on one box (csrc/mlp.hip): LMX_MLP_CFG = 0 (product: ring, QB 2), 3 / 4 (16 tokens per wave, more waves per SIMD)
for c in 0 3 5 6 3; do echo "cfg $c"; LMX_MLP_CFG=$c timeout -k 10 200 python tools/mlp_probe.py 2>&1 | grep rows; done
