"""Average rocprofv3 PMC counter values per kernel from the counter_collection CSV(s) under a directory.
    python tools/pmc_summary.py gpurun_out/pmc_dir
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            c = acc[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
    for k, cs in sorted(acc.items()):
        print(k)
        for name, (tot, cnt) in sorted(cs.items()):
            print(f"    {name:28s} avg/dispatch {tot / cnt:16.1f}   dispatches {cnt}")


if __name__ == "__main__":
    main(sys.argv[1])
