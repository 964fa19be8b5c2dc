set -e
python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r02j_parity.log 2>&1 || { tail -30 gpurun_out/r02j_parity.log; exit 1; }
tail -3 gpurun_out/r02j_parity.log
bash tools/gpu_ab.sh "$@"
