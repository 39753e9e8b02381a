"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: mean counter value per dispatch."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                k = (row.get("Kernel_Name", "?")[:90], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0) or 0)
                agg[k][1] += 1
        print("==", f)
        for (kn, cn), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
            print("%-90s %-12s dispatches=%6d mean=%.6g sum=%.6g" % (kn, cn, n, s / n, s))
