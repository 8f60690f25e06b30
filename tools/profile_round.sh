#!/bin/bash
# Profiles of the benched configuration (run on the GPU box): rocprofv3 kernel stats + four PMC passes of the SAME command,
# serialized on one stream (LMX_SERIAL=1) so that a kernel's duration is its own.  usage: bash tools/profile_round.sh r03
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LMX_SERIAL=1
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-roofline --no-per-config"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $CMD > $out/stats.log 2>&1
echo "stats rc $?"
python3 tools/kernel_stats_summary.py $out/stats "LMX_SERIAL=1 rocprofv3 --kernel-trace --stats -- python3 $CMD" > $out/${tag}_bench_kernel_stats.txt 2>&1
head -30 $out/${tag}_bench_kernel_stats.txt
bash tools/pmc.sh $out/pmc $CMD
python3 tools/pmc_summary.py $out/pmc $out/${tag}_pmc_summary.json > $out/${tag}_pmc_summary.txt 2>&1
head -20 $out/${tag}_pmc_summary.txt
