set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r02m_tests.log 2>&1 || { tail -30 gpurun_out/r02m_tests.log; exit 1; }
tail -2 gpurun_out/r02m_tests.log
python tools/gpu_nz_sweep.py
echo NO_HALF
CLIMA_HIP_NO_HALF=1 python tools/gpu_nz_sweep.py
