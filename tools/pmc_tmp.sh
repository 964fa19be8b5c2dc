cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
pargs="--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d gpurun_out/r02k_pmc_ICACHE -o run -- python3 bench.py $pargs > gpurun_out/r02k_pmc_ICACHE.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY -d gpurun_out/r02k_pmc_TCINST -o run -- python3 bench.py $pargs > gpurun_out/r02k_pmc_TCINST.log 2>&1 || exit 1
python3 tools/pmc_summary.py r02k_pmc gpurun_out/r02k_pmc_summary.md "icache"
