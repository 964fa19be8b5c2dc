set -o pipefail
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf $o/green_pmc_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $o/green_pmc_$i -o run -- python3 tools/gpu_ir_batch.py 200 > $o/green_pmc_$i.log 2>&1; echo "pass $i rc $?"
  python3 - $o/green_pmc_$i/run_counter_collection.csv <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    kn = r["Kernel_Name"]
    if "green_accum_far" in kn or "green_accum_mixed" in kn or "green_factor" in kn:
        acc[kn.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
