#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per kernel NAME (template arguments kept): launches and mean counter value per launch of
every kernel whose name contains one of the substrings.
usage: summarize_pmc_by_name.py <out.csv> <header comment> <substr,substr,...> <pass_dir> [<pass_dir> ...]"""
import collections, csv, glob, os, re, sys
out, header, subs = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
lines = []
for d in sys.argv[4:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(s in r["Kernel_Name"] for s in subs):
                name = re.sub(r"\(anonymous namespace\)::|lslam::|void ", "", r["Kernel_Name"]).split("(")[0]
                acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            lines.append((os.path.basename(os.path.normpath(d)), k, c, len(v), sum(v) / len(v), sum(v)))
with open(out, "w") as fo:
    fo.write("# " + header + "\n")
    fo.write("pass,kernel,counter,launches,mean_per_launch,total\n")
    for l in lines:
        fo.write('%s,"%s",%s,%d,%g,%g\n' % l)
print(open(out).read())
