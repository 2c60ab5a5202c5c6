#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: mean counter value per launch of one kernel.

usage: summarize_pmc.py <kernel-substring> <out.csv> <pass_dir> [<pass_dir> ...]
Launches of the kernel with the largest grid are averaged.  For bench.py's sweep that is EVERY launch of the
profiled run (a converged scan's blocks exit early, the grid does not shrink), so the figures are the mean over all
launches -- the same population as roofline.avg_kernel_ms and rocprofv3's per-kernel average.  FETCH_SIZE/WRITE_SIZE stay in KiB as reported;
bench.py applies the gfx950 correction (2 x FETCH_SIZE) from MI355X_MICROARCH.md."""
import csv, glob, os, sys, collections

def main():
    kern, out = sys.argv[1], sys.argv[2]
    lines = []
    kname = None
    for d in sys.argv[3:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
            if not rows:
                continue
            gmax = max(int(r["Grid_Size"]) for r in rows)
            rows = [r for r in rows if int(r["Grid_Size"]) == gmax]
            kname = rows[0]["Kernel_Name"]
            acc = collections.defaultdict(list)
            for r in rows:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c in sorted(acc):
                lines.append((os.path.basename(os.path.normpath(d)), c, len(acc[c]), sum(acc[c]) / len(acc[c]), gmax))
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --headline-only --steps 2 --warmup 1 "
                 "(one pass per counter set)\n")
        fo.write("# kernel: %s ; mean over the launches with the largest grid (for the bench's sweep: all of them); FETCH_SIZE/WRITE_SIZE in KiB as reported\n" % kname)
        fo.write("pass,counter,launches,mean_per_launch,grid_size\n")
        for l in lines:
            fo.write("%s,%s,%d,%g,%d\n" % l)
    print(open(out).read())

if __name__ == "__main__":
    main()
