#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: mean counter value per launch of one kernel.

usage: summarize_pmc.py <kernel-substring>[,<companion-substring>...] <out.csv> <pass_dir> [<pass_dir> ...]
EVERY launch of the kernel in the profiled run is averaged -- the same population as roofline.avg_kernel_ms and rocprofv3's
per-kernel average.  (Until round 5 only the launches with the largest grid were: every sweep of the bench launched the full
grid.  Since a batch's later sweeps are launched over the running scans' workgroups only, that filter kept the first sweep
of every step and added every sweep's second pass to it.)  Companion kernels (the second
pass of the certificate sweep: sweep_queue_kernel, cert_plan_kernel) have their counters ADDED before the division: the mean is
per sweep = per launch of the first kernel, the unit bench.py times with its HIP events.  FETCH_SIZE/WRITE_SIZE stay in KiB
as reported; bench.py applies the gfx950 correction (2 x FETCH_SIZE) from MI355X_MICROARCH.md."""
import csv, glob, os, sys, collections

def main():
    kerns, out = sys.argv[1].split(","), sys.argv[2]
    kern, companions = kerns[0], kerns[1:]
    lines = []
    kname = None
    for d in sys.argv[3:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            allrows = list(csv.DictReader(open(f)))
            rows = [r for r in allrows if kern in r["Kernel_Name"]]
            if not rows:
                continue
            gmax = max(int(r["Grid_Size"]) for r in rows)
            kname = " + ".join(sorted({r["Kernel_Name"].split("(")[0] for r in rows}))
            acc = collections.defaultdict(list)
            for r in rows:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            extra = collections.defaultdict(float)
            for r in allrows:
                if any(c in r["Kernel_Name"] for c in companions):
                    extra[r["Counter_Name"]] += float(r["Counter_Value"])
            if extra:
                kname += " (+ per sweep: " + ", ".join(companions) + ")"
            for c in sorted(acc):
                lines.append((os.path.basename(os.path.normpath(d)), c, len(acc[c]), (sum(acc[c]) + extra.get(c, 0.0)) / len(acc[c]), gmax))
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --headline-only --steps 6 --warmup 1 "
                 "(one pass per counter set)\n")
        fo.write("# kernel: %s ; mean over all launches of the kernel in the run; FETCH_SIZE/WRITE_SIZE in KiB as reported\n" % kname)
        fo.write("pass,counter,launches,mean_per_launch,grid_size\n")
        for l in lines:
            fo.write("%s,%s,%d,%g,%d\n" % l)
    print(open(out).read())

if __name__ == "__main__":
    main()
