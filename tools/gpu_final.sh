#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/gpu_final.sh r04'): everything the round's final profiles/ come from,
# written under gpurun_out/<tag>_*.  tools/finalize_profiles.sh (run in the build container afterwards) condenses
# it into profiles/.  rocprofv3 is always called with the program itself behind `--` (python3 bench.py ...), PMC
# passes one counter set at a time and never combined with other trace domains (tools/gpu_profile.sh).
# The whole pass takes ~25 minutes and one gpurun call may run 20 (round 4 learned that at the limit, with nothing
# copied back): it is run in parts -- tools/gpu_final.sh r04 lines | profile | sweeps1 | sweeps2 -- one call each.
set -o pipefail
tag=${1:-r04}
part=${2:-all}
want() { [ "$part" = all ] || [ "$part" = "$1" ]; }
o=gpurun_out
mkdir -p $o
run() { name=$1; shift; echo "== $name: $*"; "$@" > $o/${tag}_$name 2> $o/${tag}_$name.err || { echo "FAILED: $name"; tail -5 $o/${tag}_$name.err; return 1; }; }
if want lines; then
# ---- bench lines: the driver's own command, the default command, configs 3 / 4 / 5
run bench_line_steps20.json python3 bench.py --gpus 1 --steps 20 --warmup 5 || exit 1
run bench_line_default.json python3 bench.py || exit 1
run bench_line_config3.json python3 bench.py --config 3 --no-cpu-baseline || exit 1
run bench_line_config4.json python3 bench.py --config 4 --no-cpu-baseline || exit 1
run bench_line_config5.json python3 bench.py --config 5 --no-cpu-baseline || exit 1
# ---- one rank's share of an N-GPU step rehearsed on one GPU, through the LIBRARY's own RCCL step (one-rank
#      communicator + the bin shard (0, N)): configs 2 and 5
for n in 2 4 8; do
  CLIMA_BENCH_FORCE_DIST=1 CLIMA_BENCH_FAKE_SHARD=0,$n run bench_line_fake_shard_0_of_$n.json python3 bench.py --no-cpu-baseline || exit 1
  CLIMA_BENCH_FORCE_DIST=1 CLIMA_BENCH_FAKE_SHARD=0,$n run bench_line_config5_fake_shard_0_of_$n.json python3 bench.py --config 5 --no-cpu-baseline || exit 1
done
CLIMA_BENCH_FORCE_DIST=1 run bench_line_one_rank_comm.json python3 bench.py --no-cpu-baseline || exit 1
CLIMA_BENCH_FORCE_DIST=1 CLIMA_BENCH_TORCH_ALLREDUCE=1 run bench_line_one_rank_torch.json python3 bench.py --no-cpu-baseline || exit 1
run bench_line_gpus2_refused.txt bash -c 'python3 bench.py --gpus 2 --steps 2 --warmup 1; echo "exit code $?"'   # one GPU here: the launcher's one-line refusal
fi
if want profile; then
# ---- kernel statistics + PMC passes of the default workload
bash tools/gpu_profile.sh $tag || exit 1
# ... and of configs 3 and 5 (their bench lines take traffic and issue fraction from the same JSON record)
PMC_EXTRA="--config 3" bash tools/gpu_profile.sh ${tag}c3 --config 3 --steps 100 --warmup 20 --no-cpu-baseline || exit 1
PMC_EXTRA="--config 5" bash tools/gpu_profile.sh ${tag}c5 --config 5 --steps 100 --warmup 20 --no-cpu-baseline || exit 1
fi
if want sweeps1; then
# ---- sweeps
run doubled_grid.txt python3 tools/gpu_doubled_grid.py 50 100 200 || exit 1
run adiabat_like.txt python3 tools/gpu_adiabat_like.py || exit 1
run nz_sweep.txt python3 tools/gpu_nz_sweep.py || exit 1
fi
if want sweeps2; then
run ng_sweep.txt python3 tools/gpu_ng_sweep.py || exit 1
run ir_batch.txt python3 tools/gpu_ir_batch.py 50 100 200 249 || exit 1
bash tools/gpu_green_profile.sh $tag || exit 1                           # the response form's kernels under rocprofv3 (stats, timeline, PMC)
run fortran_host.txt python3 tools/gpu_fortran_host.py || exit 1      # the drop-in call timed by a Fortran host
run fortran_like.txt python3 tools/gpu_fortran_like.py || exit 1      # ... and what pushing the public fields before every call costs
run graph_ab.txt python3 tools/gpu_graph_ab.py || exit 1              # a hipGraph replay of the call's launches beside the plain launches
# ---- per-phase stamps and the block timeline (the -DCLIMA_STAMPS build, when present)
if [ -f clima_amd/csrc/libclima_radtran_hip_stamps.so ]; then
  run stamps.txt python3 tools/gpu_stamps.py || exit 1
  run timeline.txt python3 tools/gpu_timeline.py || exit 1
fi
fi
echo "final pass ($part) done: $o/${tag}_*"
