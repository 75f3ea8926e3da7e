import csv, sys
tag = sys.argv[1]
rows = [r for r in csv.DictReader(open('gpurun_out/prof_%s/p_kernel_trace.csv' % tag))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "camera_kernel" in r["Kernel_Name"] or "primary_kernel" in r["Kernel_Name"]]
fr = rows[idx[-2]:idx[-1]]
t0 = int(fr[0]["Start_Timestamp"])
sel = range(len(fr)) if len(sys.argv) < 3 else [int(x) for x in sys.argv[2].split(",")]
for i in sel:
    r = fr[i]
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
    print("%3d %8.1f +%7.1f %s" % (i, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n))
