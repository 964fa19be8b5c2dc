#!/bin/bash
# After tools/run_final_tmp.sh on the GPU box: condense into profiles/ (run here)
python tools/pmc_summary.py r02b_pmc profiles/r02b_pmc_summary.md "PMC passes, round 2 final kernels (config 2: nz=200, nw=1000, ng=8, nzen=8)" --json profiles/r02_pmc.json config2_nz200_nzen8 | tail -1
cp gpurun_out/r02b_stats/run_kernel_stats.csv profiles/r02b_kernel_stats_bench_steps100.csv
cp gpurun_out/r02b_doubled_grid.txt gpurun_out/r02b_adiabat_like.txt gpurun_out/r02b_nz_sweep.txt gpurun_out/r02b_ng_sweep.txt profiles/
