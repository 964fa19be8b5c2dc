o=gpurun_out
rm -rf $o/green_prof
sed -i 's/^for nz, ndev in .*/for nz, ndev in ((200, 12),):/' tools/probes/green_manydev.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/green_prof -o run -- python3 tools/probes/green_manydev.py > $o/green_prof.log 2>&1; echo "rc $?"
grep -i "green" $o/green_prof/run_kernel_stats.csv | cut -c1-150
