#!/bin/bash
# PMC comparison of library builds on one box: tools/gpu_pmc_ab.sh <tag> libA.so libB.so ...  (one pass per counter set
# per build; kernel trace only).  Summaries: gpurun_out/<tag>_<lib>_<set>/ -> tools/pmc_summary.py.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
[ -f gpurun_out/counters_avail.txt ] || rocprofv3 -L > gpurun_out/counters_avail.txt 2>&1
pargs="--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-jacobian ${PMC_EXTRA}"
for lib in "$@"; do
  name=$(basename $lib .so)
  export CLIMA_HIP_LIB=$PWD/$lib
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
             "SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
    s1=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/${tag}_${name}_${s1} -o run -- python3 bench.py $pargs > gpurun_out/${tag}_${name}_${s1}.log 2>&1 || echo "pass $name $s1 failed"
  done
done
unset CLIMA_HIP_LIB
