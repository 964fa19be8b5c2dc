set -o pipefail
o=gpurun_out
python3 -m pytest tests/test_gpu_ir_green.py -x -q -m gpu > $o/green_mfma_t.log 2>&1; echo "tests rc $?"; tail -3 $o/green_mfma_t.log
python3 tools/gpu_ir_batch.py 100 200 > $o/green_mfma_on.txt 2>&1 && cut -c1-200 $o/green_mfma_on.txt
rm -rf $o/green_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/green_prof -o run -- python3 tools/gpu_ir_batch.py 200 > $o/green_prof.log 2>&1; echo "rc $?"
grep -i "green\|ir_batch" $o/green_prof/run_kernel_stats.csv | cut -c1-150
