"""Summarise rocprofv3 --pmc csv output per kernel (mean per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("rx_", "omp_batch", "demod_", "mod_kernel", "eq_demap", "pilot_ls", "mmse_", "mp_batch", "t4_", "acf_", "fine_")):
            continue
        acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} {sum(v)/len(v):16.1f}  (n={len(v)})")
