"""Condenses rocprofv3 output (gpurun_out/, scratch) into the small, tracked
summaries under profiles/ that DESIGN.md and bench.py cite.

    python tools/summarize_profiles.py stats  gpurun_out/prof_r01/bench_kernel_stats.csv  profiles/r01_bench_kernel_stats.csv
    python tools/summarize_profiles.py pmc    gpurun_out/pmc_fetch/mb_counter_collection.csv \
                                              gpurun_out/pmc_write/mb_counter_collection.csv profiles/r01_pmc_traffic.json

PMC correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced streaming read, WRITE_SIZE is exact; both are
in KiB. traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch -- except
for kernels whose reads are RANDOM ROWS (RANDOM_ROW_KERNELS): calibrated on known
byte counts (profiles/r03_fetch_calibration.txt), a random 32- or 64-byte row is one
64-byte request counted whole, so their FETCH_SIZE is taken as reported.
"""

import collections
import csv
import json
import sys


RANDOM_ROW_KERNELS = ("gather_packed_kernel", "gather_minibatch_kernel")


def short(name: str) -> str:
    name = name.replace("void ", "")
    if name.startswith("Cijk"):
        i = name.find("MT")
        return "hipBLASLt " + name[5:14] + " " + name[i : i + 16]
    return name.split("(")[0][:110]


def stats(src: str, dst: str, keep: int = 25) -> None:
    rows = list(csv.DictReader(open(src)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    picked = rows[:keep] + [r for r in rows[keep:] if "rl8::" in r["Name"] or "mlp_" in r["Name"]]
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_us", "total_ms", "percent_of_kernel_time"])
        for r in picked:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e3:.2f}",
                        f"{float(r['TotalDurationNs']) / 1e6:.3f}", f"{100 * float(r['TotalDurationNs']) / total:.3f}"])
        w.writerow(["TOTAL (all kernels)", sum(int(r["Calls"]) for r in rows), "", f"{total / 1e6:.3f}", "100"])
        # One symbol, two launch sizes (since round 4 the rollout's recording launches of 2^20 rows and the training
        # launches of 2^25 run the same forward kernel): rocprofv3's average mixes them.  From the dispatch trace beside
        # the stats file: kernels whose longest dispatch is 8x their shortest, split at the geometric mean.
        trace = src.replace("kernel_stats.csv", "kernel_trace.csv")
        try:
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(trace)):
                if "rl8::" in r["Kernel_Name"]:
                    per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            for name, us in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                if len(us) < 2 or max(us) < 8 * min(us):
                    continue
                cut = (max(us) * min(us)) ** 0.5
                for label, part in (("long", [u for u in us if u >= cut]), ("short", [u for u in us if u < cut])):
                    w.writerow([f"{name} [{label} launches]", len(part), f"{sum(part) / len(part):.2f}",
                                f"{sum(part) / 1e3:.3f}", f"{100 * sum(part) * 1e3 / total:.3f}"])
        except FileNotFoundError:
            pass
    print(f"wrote {dst}: {len(picked)} rows")


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def pmc(fetch_csv: str, write_csv: str, dst: str) -> None:
    def load(path, counter):
        agg = collections.defaultdict(list)
        grid = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and ("rl8::" in r["Kernel_Name"] or "mlp_" in r["Kernel_Name"]):
                k = short(r["Kernel_Name"])
                agg[k].append(float(r["Counter_Value"]))
                grid[k] = int(r["Grid_Size"])
        return agg, grid

    fetch, grid = load(fetch_csv, "FETCH_SIZE")
    write, _ = load(write_csv, "WRITE_SIZE")
    out = {}
    for k in fetch:
        f, w = median(fetch[k]), median(write.get(k, [0.0]))
        random_rows = any(n in k for n in RANDOM_ROW_KERNELS)
        out[k] = {
            "launches_sampled": len(fetch[k]),
            "grid_size": grid[k],
            "FETCH_SIZE_KiB_median": f,
            "WRITE_SIZE_KiB_median": w,
            "traffic_bytes_per_launch": ((1 if random_rows else 2) * f + w) * 1024,
            "note": ("FETCH_SIZE (random rows: counted whole, profiles/r03_fetch_calibration.txt) + WRITE_SIZE, KiB -> bytes"
                     if random_rows else "2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction), KiB -> bytes"),
        }
    json.dump(out, open(dst, "w"), indent=1)
    print(f"wrote {dst}: {len(out)} kernels")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
