#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 `--pmc ... --output-format csv` counter_collection.csv."""
import csv
import sys
from collections import defaultdict, OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, ctrs in acc.items():
    n = max(1, len(disp[k]))
    print("%s  (%d dispatches; per-dispatch averages)" % (k, n))
    for c, v in sorted(ctrs.items()):
        print("    %-28s %18.1f" % (c, v / n))
