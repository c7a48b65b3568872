#!/bin/bash
# instruction-cache counters of the encoder's kernels: the default line and the progressive one
cd /tmp && export TMPDIR=/tmp
for mode in default progressive; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache_$mode; rm -rf $out; mkdir -p $out
  extra=""; [ $mode = progressive ] && extra="--progressive"
  timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_IFETCH --kernel-trace -d $out -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline --no-psnr > $out/log.txt 2>&1
  echo "$mode rc=$?"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/pmc_icache_default gpurun_out/pmc_icache_progressive 2>&1 | grep -E "^==|k_transform|k_encode|k_prog2|k_compact" | tee gpurun_out/icache_summary.txt
