#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per KERNEL: for every kernel whose name matches one of the given substrings, launches,
mean counter value per launch and the total per build (sum over the launches / number of builds in the run).

usage: summarize_pmc_kernels.py <out.csv> <n_builds> <substr,substr,...> <pass_dir> [<pass_dir> ...]
FETCH_SIZE / WRITE_SIZE stay in KiB as reported (bench.py applies the gfx950 correction 2 x FETCH_SIZE)."""
import collections, csv, glob, os, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|lslam::", "", name)
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def main():
    out, n_builds, subs = sys.argv[1], int(sys.argv[2]), sys.argv[3].split(",")
    lines = []
    for d in sys.argv[4:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if any(s in r["Kernel_Name"] for s in subs):
                    acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(acc.items()):
                lines.append((os.path.basename(os.path.normpath(d)), k, c, len(v), sum(v) / len(v), sum(v) / n_builds))
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 tools/bench_treebuild.py (REPS=%d builds of the "
                 "157 k + 587 k-point surround: both trees), one pass per counter set; FETCH_SIZE / WRITE_SIZE in KiB as reported\n" % n_builds)
        fo.write("pass,kernel,counter,launches,mean_per_launch,total_per_build\n")
        for l in lines:
            fo.write("%s,%s,%s,%d,%g,%g\n" % l)
    print(open(out).read())


if __name__ == "__main__":
    main()
