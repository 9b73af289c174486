"""Summary of a tools/profile_all.sh C4 directory: kernel stats + per-launch counters of the transfer_dense kernels."""
import collections, csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:90], "calls", r["Calls"], "avg ns", r["AverageNs"])
for d in sorted(glob.glob(out + "/pmc_*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:80]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "transfer_dense" in k:
            print(k)
            for c, v in acc[k].items():
                print("   %-28s %.5g per launch (n=%d)" % (c, v / n[(k, c)], n[(k, c)]))
