#!/bin/bash
mkdir -p gpurun_out/gemm_sweep
timeout -k 10 200 python tools/gemm_sweep.py gpurun_out/gemm_sweep/sweep_default.json 2>&1 | tail -1
for v in C D H I A T U Y E; do
  LMX_GEMM2_VARIANT=$v timeout -k 10 200 python tools/gemm_sweep.py gpurun_out/gemm_sweep/sweep_$v.json 2>&1 | tail -1 || exit 1
done
python tools/gemm_sweep_report.py gpurun_out/gemm_sweep | tee gpurun_out/gemm_sweep/report.txt | head -60
