set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r02j_parity.log 2>&1 || { tail -30 gpurun_out/r02j_parity.log; exit 1; }
tail -2 gpurun_out/r02j_parity.log
bash tools/gpu_ab.sh "$@"
