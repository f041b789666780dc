"""dev aid: condense rocprofv3 outputs under gpurun_out/ into the summaries kept in profiles/."""
import csv, json, collections, sys
ROUND = "r01"
out = "profiles"
# 1. kernel stats of the bench command
rows = list(csv.DictReader(open("gpurun_out/prof_final/bench_kernel_stats.csv")))
with open(f"{out}/{ROUND}_bench_kernel_stats_final.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py  (round 1, end of round; under the\n")
    f.write("# profiler the multigrid iterations are launched kernel by kernel instead of replayed as hipGraphs: lbc.hip graphs_enabled())\n")
    f.write("kernel,calls,avg_us,total_ms,percent\n")
    for r in rows:
        f.write('"%s",%s,%.2f,%.3f,%s\n' % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3,
                                           float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
# 2. PMC passes
def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
fetch = pmc("gpurun_out/prof_fetch/f_counter_collection.csv", "FETCH_SIZE")
write = pmc("gpurun_out/prof_write/w_counter_collection.csv", "WRITE_SIZE")
cmd = "python3 bench.py --steps 3 --warmup 1 --no-rays --no-knn --no-skeleton --no-ransac --no-cpu"
for name, acc, cname in (("fetch_size", fetch, "FETCH_SIZE"), ("write_size", write, "WRITE_SIZE")):
    with open(f"{out}/{ROUND}_dbscan_pmc_{name}_final.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc {cname} -- {cmd}  (round 1, end of round)\n")
        f.write("# counter unit: KiB as reported by rocprofv3; gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)\n")
        f.write("# coalesced streaming reads (MI355X_MICROARCH.md, HBM section); these kernels read 4-8 B per lane: uncalibrated\n")
        f.write("kernel,dispatches,avg_counter_value_kib\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.2f\n' % (k, len(v), sum(v) / len(v)))
kern = {}
for scope, k in (("k_core_tiled", "pyqsm::k_core_tiled"), ("k_hook_sub", "pyqsm::k_hook_sub"), ("k_union_sub", "pyqsm::k_union_sub")):
    fk = sum(fetch[k]) / len(fetch[k]); wk = sum(write[k]) / len(write[k])
    kern[scope] = {"fetch_size_kib": fk, "write_size_kib": wk, "hbm_bytes_per_launch": (fk + wk) * 1024.0}
json.dump({"points": 1_000_000, "kernels": kern,
           "note": "FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 --pmc passes (profiles/r01_dbscan_pmc_*_final.csv), "
                   "raw counter x 1024 B; the gfx950 2x correction of FETCH_SIZE is calibrated for 16 B/lane streaming reads only "
                   "and is NOT applied (these kernels read 4-8 B per lane), so the figure is a lower bound of the read traffic"},
          open(f"{out}/{ROUND}_dbscan_traffic.json", "w"), indent=1)
print(json.dumps(kern, indent=1))
