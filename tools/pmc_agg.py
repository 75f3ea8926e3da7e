import csv, glob, collections, sys
for tag in sys.argv[1:]:
    f = glob.glob("gpurun_out/pmc_%s/**/p_counter_collection.csv" % tag, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[k] += 1
    for k, v in agg.items():
        if any(s in k for s in ("group_chain", "trace_kernel", "primary")):
            print("%-36s n=%3d " % (k, calls[k]) + "  ".join("%s=%.4g" % (c.replace("_sum", ""), x / calls[k]) for c, x in v.items()))
