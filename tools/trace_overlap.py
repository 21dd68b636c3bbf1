"""Does post_select_kernel overlap the forward's kernels?  Reads a rocprofv3 kernel-trace csv (gpurun_out/prof_bs1/*_kernel_trace.csv):
for the last few post_select_kernel dispatches prints which kernels ran during them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r["Kernel_Name"].split("(")[0][-60:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "post_select_kernel" in r["Kernel_Name"]][-3:]
for s in sel:
    a, b = int(s["Start_Timestamp"]), int(s["End_Timestamp"])
    print("select %d ns, stream/queue %s" % (b - a, s.get("Queue_Id")))
    for r in rows:
        x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if y > a and x < b and r is not s:
            print("   overlaps %-60s %7d ns  queue %s  (from %+d)" % (name(r), y - x, r.get("Queue_Id"), x - a))
    # neighbours in time
    i = rows.index(s)
    for r in rows[max(0, i - 3):i + 4]:
        print("   near %-60s start %+8d end %+8d queue %s" % (name(r), int(r["Start_Timestamp"]) - a, int(r["End_Timestamp"]) - a, r.get("Queue_Id")))
