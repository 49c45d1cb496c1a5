"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into
profiles/pmc_traffic.json (memory-side bytes per launch per kernel), applying the gfx950
correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 128-B requests at 64 B, so the
read side is doubled; WRITE_SIZE is exact.  The doubling is the guide's for wide coalesced
reads and was calibrated for this library's other pattern -- one random 128-byte record per
lane read as 7 x 16 B, k_accumulate's gather -- by tools/ubench_gather.hip: raw counter =
0.49 x the known bytes there too (profiles/r02_fetch_calibration.txt).  Both counters are in KiB.
    python tools/pmc_summary.py <fetch.csv> <write.csv> [out.json]
"""
import collections, csv, json, re, sys

def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"curdle::k_(\w+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"].split("(")[0]
        agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write)):
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"fetch_size_kib_raw": f_kib, "write_size_kib": w_kib,
              "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024)}
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
import subprocess, datetime
try:
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = ""
out["_measured"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 3`, "
                    + datetime.date.today().isoformat() + (", tree at " + head if head else "")
                    + "; read side x2 (calibrated: profiles/r02_fetch_calibration.txt)")
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    if k.startswith("_"):
        continue
    print(f"{k:24s} fetch(raw) {v['fetch_size_kib_raw']/1024:10.1f} MiB  write {v['write_size_kib']/1024:10.1f} MiB  hbm {v['hbm_bytes_per_launch']/2**20:10.1f} MiB")
