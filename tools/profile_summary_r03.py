"""dev aid: condense the rocprofv3 outputs of tools/profile_r03.sh into the summaries kept under
profiles/ (tracked): kernel stats, FETCH_SIZE / WRITE_SIZE passes (DBSCAN step, kNN, Laplacian
build, contraction solve), SQ counter passes, and r03_traffic.json, which bench.py reads for its
`roofline.traffic` fields."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROUND = "r03"
SRC = sys.argv[1] if len(sys.argv) > 1 else "/tmp/prof_r03"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles"
os.makedirs(OUT, exist_ok=True)


def find(sub, suffix):
    hits = sorted(glob.glob(f"{SRC}/{sub}/**/*{suffix}", recursive=True))
    return hits[-1] if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")


def plain(name):
    """kernel name without namespace and template arguments: the key bench.py looks up"""
    return re.sub(r"<.*>$", "", short(name).replace("pyqsm::", ""))


def last_json(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


def stats(sub, out_name, header):
    path = find(sub, "kernel_stats.csv")
    if not path:
        return {}
    rows = sorted(csv.DictReader(open(path)), key=lambda r: float(r["AverageNs"]))   # heavy instantiation wins below
    with open(f"{OUT}/{ROUND}_{out_name}.csv", "w") as f:
        f.write(header)
        f.write("kernel,calls,avg_us,total_ms,percent\n")
        for r in rows:
            f.write('"%s",%s,%.2f,%.3f,%s\n' % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                  float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    return {plain(r["Name"]): {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                               "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in rows}


def pmc(sub, counter):
    """{kernel (plain name): [per-dispatch counter values]}"""
    path = find(sub, "counter_collection.csv")
    acc = collections.defaultdict(list)
    if path:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[plain(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def traffic(fetch_sub, write_sub, units, csv_name, cmd, level0=()):
    """Per kernel: average FETCH_SIZE / WRITE_SIZE per dispatch (KiB), HBM bytes per launch =
    (2 x FETCH + WRITE) x 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request), launches per
    unit of work (step / build / solve) and bytes per unit. For kernels in `level0` only the
    dispatches with at least half the kernel's largest counter value are kept (the finest level of
    the multigrid: the coarse levels run the same kernels on smaller operators)."""
    fetch, write = pmc(fetch_sub, "FETCH_SIZE"), pmc(write_sub, "WRITE_SIZE")
    for cname, acc in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        with open(f"{OUT}/{ROUND}_{csv_name}_pmc_{cname.lower()}.csv", "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --pmc {cname} -- {cmd}  (round 3)\n")
            f.write("# counter unit: KiB as reported by rocprofv3; on gfx950 FETCH_SIZE reports half the bytes read\n")
            f.write("# (MI355X_MICROARCH.md, HBM section): double it before comparing with a byte count; WRITE_SIZE is exact\n")
            f.write("kernel,dispatches,avg_counter_value_kib,sum_counter_value_kib\n")
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                f.write('"%s",%d,%.2f,%.2f\n' % (k, len(v), sum(v) / len(v), sum(v)))
    kern, total = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_") and "fillBuffer" not in k and "copyBuffer" not in k:
            continue
        fv, wv = fetch.get(k, []), write.get(k, [])
        if k in level0 and fv:
            cut = 0.5 * max(fv)
            keep = [i for i, x in enumerate(fv) if x >= cut]
            wv = [wv[i] for i in keep if i < len(wv)]
            fv = [fv[i] for i in keep]
        nf, nw = max(len(fv), 1), max(len(wv), 1)
        fk, wk = sum(fv) / nf, sum(wv) / nw
        per_unit = max(len(fv), len(wv)) / float(units)
        kern[k] = {"fetch_size_kib": fk, "write_size_kib": wk, "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                   "launches_per_unit": per_unit}
        total += (2.0 * fk + wk) * 1024.0 * per_unit
    return kern, total


def sq(sub, out_name, cmdline, keep):
    path = find(sub, "counter_collection.csv")
    if not path:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = []
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
    names.sort()
    out = {}
    with open(f"{OUT}/{ROUND}_{out_name}.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc {' '.join(names)} -- {cmdline}  (round 3)\n")
        f.write("# per-dispatch averages; SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over "
                "waves (MI355X_MICROARCH.md)\n")
        f.write("kernel,dispatches," + ",".join(names) + "\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
            if keep and not any(s in k for s in keep):
                continue
            n = max(len(x) for x in v.values())
            f.write('"%s",%d,' % (k, n) + ",".join("%.0f" % (sum(v[c]) / len(v[c])) if v.get(c) else "" for c in names) + "\n")
            rec = {c: sum(v[c]) / len(v[c]) for c in names if v.get(c)}
            if plain(k) not in out or rec.get("SQ_WAVE_CYCLES", 0) > out[plain(k)].get("SQ_WAVE_CYCLES", 0):
                out[plain(k)] = rec          # template instantiations share a plain name: keep the heavy one
    return out


DB = "python3 bench.py --steps 5 --warmup 1 --no-rays --no-knn --no-skeleton --no-ransac --no-cpu"
KN = "python3 bench.py --steps 2 --warmup 1 --no-rays --no-skeleton --no-ransac --no-cpu"

# 1. kernel stats of the whole default bench run
bench_stats = stats("bench", "bench_kernel_stats",
      "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu --no-config5  (round 3;\n"
      "# hipGraph replays are ON under the profiler)\n")
bj = last_json(f"{SRC}/bench.json")
if bj:
    open(f"{OUT}/{ROUND}_bench_under_rocprof.json", "w").write(json.dumps(bj) + "\n")
lap_stats = stats("lap_stats", "laplacian_kernel_stats",
                  "# rocprofv3 --kernel-trace --stats -- python3 tools/prof_skel.py lap  (three builds, 1 M points)\n")
sol_stats = stats("sol_stats", "solve_kernel_stats",
                  "# rocprofv3 --kernel-trace --stats -- python3 tools/prof_skel.py solve  (one build + the first contraction solve)\n")

# 2. HBM traffic
out = {"points": 1_000_000,
       "note": "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes "
               f"(profiles/{ROUND}_*_pmc_*.csv): gfx950 FETCH_SIZE reports half the bytes read (MI355X_MICROARCH.md, "
               "HBM), calibrated on this code's own k_bbox (24 B per point read: FETCH_SIZE says 12); WRITE_SIZE is "
               "exact (k_bk_scatter: one 32-byte record per point). launches_per_unit = dispatches seen / units of work in the pass; "
               "hbm_bytes_per_unit = sum over the kernels."}
# the DBSCAN pass runs warmup + steps + one profiling-level-2 call = 7 clusterings
k, tot = traffic("fetch", "write", 7, "dbscan", DB)
out["dbscan"] = {"unit": "one clustering of the 1 M-point forest (dbscan step)", "units_in_pass": 7, "kernels": k,
                 "hbm_bytes_per_unit": tot}
if "k_bbox" in k and "k_bk_scatter" in k:
    out["calibration"] = {
        "k_bbox (reads the 24 B/point AoS cloud once)": {"actual_read_bytes": 24.0e6,
                                                         "FETCH_SIZE_bytes": k["k_bbox"]["fetch_size_kib"] * 1024},
        "k_bk_scatter (writes one 32-byte record per point, block counters aside)": {
            "actual_write_bytes": 32.0e6, "WRITE_SIZE_bytes": k["k_bk_scatter"]["write_size_kib"] * 1024}}
lapj, solj = last_json(f"{SRC}/lap_fetch.json"), last_json(f"{SRC}/sol_fetch.json")
if lapj:
    k, tot = traffic("lap_fetch", "lap_write", lapj["builds"], "laplacian", "python3 tools/prof_skel.py lap")
    out["laplacian"] = {"unit": "one point-cloud Laplacian build (kNN + fans + cover + flips + assembly), 1 M points",
                        "units_in_pass": lapj["builds"], "nnz": lapj["nnz"], "kernels": k, "hbm_bytes_per_unit": tot,
                        "kernel_ms_per_unit": {q: v["total_ms"] / lapj["builds"] for q, v in lap_stats.items()
                                               if q.startswith("k_")}}
if solj:
    l0 = ("k_bspmv_f", "k_down", "k_up_ap", "k_update_r_f", "k_direction_f", "k_restrict")
    k, tot = traffic("sol_fetch", "sol_write", 1, "solve", "python3 tools/prof_skel.py solve", level0=l0)
    out["solve"] = {"unit": "level-0 (1 M rows) dispatches of the first contraction solve; coarse-level dispatches of the "
                            "same kernels are dropped (counter below half the kernel's maximum)",
                    "iters": solj.get("iters"), "nnz": solj["nnz"],
                    "kernels": {q: v for q, v in k.items() if q in l0}}
k, tot = traffic("knn_fetch", "knn_write", 4, "knn", KN)   # 1 untimed + 2 timed + 1 after the stray run ... see bench.py
out["knn"] = {"unit": "per dispatch (bench.py's kNN section, k = 20)", "kernels": {q: v for q, v in k.items() if "knn" in q}}

# 3. SQ counter passes
sq("sq_dbscan", "dbscan_sq_counters", DB, None)
knn_sq = sq("sq_knn", "knn_sq_counters", KN, ["k_knn"])
reg = knn_sq.get("k_knn_reg")
dur = bench_stats.get("k_knn_reg", {}).get("avg_us")
if reg and dur:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all waves: x 4 / 1024 SIMDs = cycles a SIMD's vector
    # pipe was issuing; the launch lasts dur x 2.4 GHz cycles (42 %)
    cyc = dur * 1e-6 * 2.4e9
    out["knn"]["k_knn_reg_sq"] = reg
    out["knn"]["k_knn_reg_avg_us"] = dur
    out["knn"]["valu_busy"] = reg["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
    out["knn"]["wait_inst_any"] = reg["SQ_WAIT_INST_ANY"] / reg["SQ_WAVE_CYCLES"]
    out["knn"]["wait_any"] = reg["SQ_WAIT_ANY"] / reg["SQ_WAVE_CYCLES"]
json.dump(out, open(f"{OUT}/{ROUND}_traffic.json", "w"), indent=1)

# 4. graphs under the profiler
lines = []
for mode in ("jacobi", "amg"):
    try:
        log = open(f"{SRC}/graph_{mode}.log").read()
    except OSError:
        log = ""
    lines.append(f"# rocprofv3 --kernel-trace --stats -- python3 tools/graph_under_rocprof.py {mode}"
                 + ("   (PYQSM_AMG_GRAPH=1 exported in the shell before rocprofv3)" if mode == "amg" else ""))
    lines += [q for q in log.splitlines() if "solve returned" in q or "laplacian built" in q] or ["(no output)"]
open(f"{OUT}/{ROUND}_graph_under_rocprof.txt", "w").write("\n".join(lines) + "\n")
print(json.dumps({q: (v.get("hbm_bytes_per_unit") if isinstance(v, dict) else None) for q, v in out.items()}, indent=1))
