"""HBM bytes per launch of the chain kernels from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh.
usage: python tools/pmc_traffic.py <pmc-dir> [frames-per-step] > profiles/roundN/traffic.json
FETCH_SIZE / WRITE_SIZE are in KB (1024 B); FETCH_SIZE is doubled (gfx950 counts half of wide streaming reads,
MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections, csv, glob, json, sys
d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20480      # bench.py default
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/tcc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        for key in ("rx_pilot_omp_kernel", "rx_pilot_kernel", "omp_batch_kernel", "rx_symbols_wave_kernel", "rx_symbols_kernel"):
            if key + "<" in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out, tot = {}, 0.0
for k, v in acc.items():
    f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
    w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    b = (2 * f + w) * 1024
    out[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": b}
    tot += b
out["chain_bytes_per_step"] = tot
out["frames"] = frames
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc.sh) of `bench.py --no-cpu`, "
               "%d frames of config M fp32;" % frames + " FETCH_SIZE doubled for gfx950; unit KB = 1024 B")
print(json.dumps(out, indent=1))
