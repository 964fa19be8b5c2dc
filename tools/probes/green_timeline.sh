set -o pipefail
o=gpurun_out
rm -rf $o/green_prof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o/green_prof -o run -- python3 tools/gpu_ir_batch.py 200 > $o/green_prof.log 2>&1; echo "rc $?"
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/green_prof/run_kernel_trace.csv")))
rows = [r for r in rows if "green" in r["Kernel_Name"] or "ir_batch" in r["Kernel_Name"] or "integrate" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last response-form batch: from the last k_green_factor on
idx = max(i for i, r in enumerate(rows) if "k_green_factor" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-60s queue %s  %8.1f -> %8.1f us" % (r["Kernel_Name"][:60], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
