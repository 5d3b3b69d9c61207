"""Summarise rocprofv3 --pmc output: per kernel, the mean counter value per launch.

    python tools/pmc_summary.py TAG DIR [DIR ...]

Walks each DIR for *counter_collection.csv (rocprofv3 --output-format csv) and
prints one line per kernel: ``TAG <kernel> {counter: mean} launches= n``.
Kernel names are cut at the first '(' so template arguments stay visible.
"""

import csv
import os
import sys
from collections import defaultdict


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
    for d in dirs:
        for root, _, files in os.walk(d):
            for f in files:
                if not f.endswith("counter_collection.csv"):
                    continue
                with open(os.path.join(root, f), newline="") as fh:
                    for row in csv.DictReader(fh):
                        k = row["Kernel_Name"].split("(")[0]
                        acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for k in sorted(acc):
        means = {c: round(sum(v.values()) / len(v), 1) for c, v in sorted(acc[k].items())}
        n = max(len(v) for v in acc[k].values())
        print(tag, k, means, "launches=", n)


if __name__ == "__main__":
    main()
