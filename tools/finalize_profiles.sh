#!/bin/bash
# After `gpurun -- 'bash tools/gpu_final.sh r04'`: condense gpurun_out/r04_* into profiles/ (run here).  ONE run,
# ONE profiles commit at the end of the round.
tag=${1:-r04}
python3 tools/pmc_summary.py ${tag}_pmc profiles/${tag}_pmc_summary.md "PMC passes, round 4 final kernels (config 2: nz=200, nw=1000, ng=8, nzen=8)" --json profiles/${tag}_pmc.json config2_nz200_nzen8 | tail -1
cp gpurun_out/${tag}_stats/run_kernel_stats.csv profiles/${tag}_kernel_stats_bench_steps100.csv
if [ -d gpurun_out/${tag}c3_pmc_FETCH_SIZE ]; then
  python3 tools/pmc_summary.py ${tag}c3_pmc profiles/${tag}_pmc_summary_config3.md "PMC passes, round 4 final kernels (config 3: EarlyMars, nz=200, nw=1000, ng=8, nzen=4)" --json profiles/${tag}_pmc.json config3_nz200_nzen4 | tail -1
  cp gpurun_out/${tag}c3_stats/run_kernel_stats.csv profiles/${tag}_kernel_stats_config3_steps100.csv
fi
if [ -d gpurun_out/${tag}c5_pmc_FETCH_SIZE ]; then
  python3 tools/pmc_summary.py ${tag}c5_pmc profiles/${tag}_pmc_summary_config5.md "PMC passes, round 4 final kernels (config 5: one 500-layer column, nw=1000, ng=8, nzen=8)" --json profiles/${tag}_pmc.json config5_nz500_nzen8 | tail -1
  cp gpurun_out/${tag}c5_stats/run_kernel_stats.csv profiles/${tag}_kernel_stats_config5_steps100.csv
fi
for f in gpurun_out/${tag}_bench_line_*.json gpurun_out/${tag}_doubled_grid.txt gpurun_out/${tag}_adiabat_like.txt gpurun_out/${tag}_nz_sweep.txt \
         gpurun_out/${tag}_ng_sweep.txt gpurun_out/${tag}_ir_batch.txt gpurun_out/${tag}_fortran_host.txt gpurun_out/${tag}_fortran_like.txt gpurun_out/${tag}_graph_ab.txt gpurun_out/${tag}_stamps.txt gpurun_out/${tag}_timeline.txt gpurun_out/${tag}_bench_line_gpus2_refused.txt; do
  [ -s "$f" ] && grep -v "amdgpu.ids" "$f" > profiles/$(basename "$f")
done
[ -s gpurun_out/${tag}_green_stats/run_kernel_stats.csv ] && grep -i "green\|ir_batch\|Name" gpurun_out/${tag}_green_stats/run_kernel_stats.csv > profiles/${tag}_green_kernel_stats.csv
for f in gpurun_out/${tag}_green_timeline.txt gpurun_out/${tag}_green_pmc.txt; do [ -s "$f" ] && cp "$f" profiles/; done
ls profiles | grep "^${tag}_"
