"""Summarise a rocprofv3 --pmc CSV run: per kernel name, dispatch count and mean counter value."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root, counter):
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", root)
        return
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row.get("Kernel_Name", "?")[:90]
                a = acc[k]
                a[0] += 1
                a[1] += float(row.get("Counter_Value", 0) or 0)
    print(f"{counter}: kernel, dispatches, mean value per dispatch (FETCH_SIZE/WRITE_SIZE are in KiB; gfx950: double FETCH_SIZE)")
    for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:90s} {n:8d} {tot / max(n, 1):14.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
