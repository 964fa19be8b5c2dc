#!/bin/bash
# Run on the GPU box (through gpurun): the response form of radtran_radiate_ir_batch (ir_green.inc) under rocprofv3 --
# kernel statistics of tools/gpu_ir_batch.py at 402 layers x 403 columns, the kernel timeline of one batch, then one PMC
# pass per counter set (never combined with other trace domains).  Output: gpurun_out/<tag>_green_*.
#   tools/gpu_green_profile.sh <tag>
set -o pipefail
tag=${1:-r04}
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf $o/${tag}_green_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_green_stats -o run -- python3 tools/gpu_ir_batch.py 200 > $o/${tag}_green_stats.log 2>&1 || exit 1
python3 - $o/${tag}_green_stats/run_kernel_trace.csv > $o/${tag}_green_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "green" in r["Kernel_Name"] or "ir_batch" in r["Kernel_Name"] or "integrate_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_green_factor" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
print("kernels of the last response-form batch (402 layers x 403 columns), us from the first one's start")
for r in rows[idx:idx + 10]:
    if "integrate_one" in r["Kernel_Name"]: break
    print("%-64s %8.1f -> %8.1f" % (r["Kernel_Name"][:64], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $o/${tag}_green_pmc_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $o/${tag}_green_pmc_$i -o run -- python3 tools/gpu_ir_batch.py 200 > $o/${tag}_green_pmc_$i.log 2>&1 || exit 1
done
python3 - $o $tag > $o/${tag}_green_pmc.txt <<'PY'
import csv, sys, collections, glob
o, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("%s/%s_green_pmc_*/run_counter_collection.csv" % (o, tag))):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_green_" in kn:
            acc[kn.split("(")[0].replace("clima::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("PMC counters of the response form's kernels, averages per launch (tools/gpu_ir_batch.py 200: 402 layers x 403 columns, 600 IR bins x 8 g-points)")
print("(SQ_* summed over the chip; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, uncorrected)")
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-32s %16.0f" % (c, sum(v) / len(v)))
PY
echo "green profile done: $o/${tag}_green_*"
