#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace statistics of the default bench command,
# then ONE PMC pass per counter set (counters are never combined with other trace domains), everything
# under gpurun_out/<tag>_*.  tools/pmc_summary.py condenses the passes into profiles/.
#   tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
tag=${1:-r02}; shift
args=${@:---steps 100 --warmup 20 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o run -- python3 bench.py $args > gpurun_out/${tag}_stats.log 2>&1 || exit 1
pargs="--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline ${PMC_EXTRA}"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/${tag}_pmc_${name} -o run -- python3 bench.py $pargs > gpurun_out/${tag}_pmc_${name}.log 2>&1 || exit 1
done
echo "profile passes done: gpurun_out/${tag}_*"
