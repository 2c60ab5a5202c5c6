#!/usr/bin/env python3
"""Per-dispatch counter values of the kernels whose name contains <substring>, in dispatch order, from rocprofv3 --pmc
passes (csv):   python3 tools/pmc_per_dispatch.py <substring> <pass_dir> [<pass_dir> ...] [-n last_N]"""
import csv, glob, os, sys, collections
args = sys.argv[1:]
n = 12
if "-n" in args:
    i = args.index("-n"); n = int(args[i + 1]); del args[i:i + 2]
kern, dirs = args[0], args[1:]
for d in dirs:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        disp = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(disp)[-n:]
        names = sorted({c for i in ids for c in disp[i]})
        print("# %s: last %d dispatches of *%s*" % (d, len(ids), kern))
        print("dispatch " + " ".join("%16s" % c[:16] for c in names))
        for i in ids:
            print("%8d " % i + " ".join("%16.4g" % disp[i].get(c, float("nan")) for c in names))
