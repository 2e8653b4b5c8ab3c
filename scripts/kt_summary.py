#!/usr/bin/env python3
"""Per-kernel summary of one `rocprofv3 --kernel-trace` run of the DEFAULT bench command (the streamed shard), from the per-dispatch
trace: calls, total, average, min, max -- and the average over the CHUNK launches alone.  The default command also makes ONE
whole-shard launch of every kernel before the timed steps (the resident-tile rows every streamed step is checked against); that
launch is the longest dispatch of each kernel and is left out of `average_ns_chunk_launches`, the figure bench.py's
roofline.avg_launch_ms (HIP events over the timed steps' chunk launches) is to be compared with.
usage: python scripts/kt_summary.py <rocprof output dir> <out.csv>"""
import collections
import csv
import glob
import os
import sys

d, out = sys.argv[1], sys.argv[2]
per = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "nfx" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out, "w") as fh:
    fh.write("kernel,calls,total_ns,average_ns,min_ns,max_ns,average_ns_chunk_launches\n")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        rest = sorted(v)[:-1] if len(v) > 1 else v
        fh.write(f"\"{k}\",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)},{sum(rest) / len(rest):.1f}\n")
print(open(out).read())
