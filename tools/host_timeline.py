"""Timeline of the last step(s) of encode_p --bench under rocprofv3 (tools/host_prof.sh): copies and kernels, in us
from the step's first event.  python tools/host_timeline.py [dir] [nevents]"""
import csv, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/host_prof"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
for r in csv.DictReader(open(f"{d}/hp_memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), r["Stream_Id"]))
for r in csv.DictReader(open(f"{d}/hp_kernel_trace.csv")):
    name = r["Kernel_Name"].split("(")[0].replace("void x265amd::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Stream_Id"]))
ev.sort()
ev = ev[-n:]
t0 = ev[0][0]
for s, e, name, st in ev:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  s{st}  {name}")
