"""Per-kernel statistics out of rocprofv3 --kernel-trace CSVs, with the launches of one kernel kept apart by what they are (the 24-image launches
hid inside the single-image kernel's row): rows are (run, kernel, grid.y, stream class). Stream class: "launch stream" = the stream the caller launched on;
"library streams, overlapping" = fri_hip_time_transform_quant_streams_dev's launches dealt over two streams (their durations overlap: a period is reported for
them, not these numbers). For the bench's trace a row `timed region` holds the K single-image launches in front of the first multi-image launch = the steps
bench.py timed (the launches before them: spin-up and warm-up).
    python3 tools/kernel_stats_from_traces.py gpurun_out/<tag> [run ...] > kernel_stats.csv"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

root = sys.argv[1]
runs = sys.argv[2:] or sorted(d for d in os.listdir(root) if d.startswith("trace_") and os.path.isdir(os.path.join(root, d)))
w = csv.writer(sys.stdout)
w.writerow(["run", "Name", "GridY", "Streams", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev"])


def emit(run, name, gy, cls, d):
    w.writerow([run, name[:130], gy, cls, len(d), f"{statistics.mean(d):.1f}", min(d), max(d), f"{statistics.pstdev(d):.1f}"])


for run in runs:
    for f in glob.glob(os.path.join(root, run, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "fri::" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        main_stream = defaultdict(lambda: defaultdict(int))
        for r in rows:
            main_stream[r["Kernel_Name"]][r["Stream_Id"]] += 1
        groups = defaultdict(list)
        for r in rows:
            name = r["Kernel_Name"]
            top = max(main_stream[name], key=main_stream[name].get)
            cls = "launch stream" if r["Stream_Id"] == top else "library streams, overlapping"
            groups[(name, r["Grid_Size_Y"], cls)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (name, gy, cls), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            emit(run, name, gy, cls, d)
        if run.startswith("trace_bench"):  # the timed steps of bench.py: the single-image launches of the product instance right in front of the first 24-image launch
            prod = [r for r in rows if "fwd_transform_quant_kernel<1" in r["Kernel_Name"] and r["Kernel_Name"].split(">(")[0].endswith("false")]
            first_batch = next((i for i, r in enumerate(prod) if r["Grid_Size_Y"] != "1"), None)
            if first_batch:
                k = 400 if first_batch >= 400 else first_batch
                d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in prod[first_batch - k:first_batch]]
                emit(run, "timed region of bench.py: the last %d single-image launches before the batch form = its K steps | " % k + prod[0]["Kernel_Name"], "1", "launch stream", d)
