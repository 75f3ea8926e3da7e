import csv, glob, collections, sys
# per counter: value of the single longest-running trace_kernel dispatch (the leaf-ray launch) and of all dispatches
for tag in sys.argv[1:]:
    f = glob.glob("gpurun_out/pmc_%s/**/p_counter_collection.csv" % tag, recursive=True)
    disp = collections.defaultdict(dict); name = {}; dur = {}
    for r in csv.DictReader(open(f[0])):
        d = r["Dispatch_Id"]; disp[d][r["Counter_Name"]] = float(r["Counter_Value"]); name[d] = r["Kernel_Name"]
        dur[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tk = [d for d in disp if "::trace_kernel" in name[d]]
    big = max(tk, key=lambda d: dur[d])
    print(tag, "longest trace_kernel dispatch: %.1f us" % (dur[big] / 1e3), {k: "%.4g" % v for k, v in disp[big].items()})
