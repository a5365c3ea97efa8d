"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a bench.py run into per-launch HBM traffic of one kernel.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/collect_traffic.py out/f out/w linear_bf16 profiles/r01/vit_traffic.json

Units and gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950, so it is doubled; WRITE_SIZE is exact for
16-byte streaming stores.  Infinity-Cache hits are included (the counters sit on the L2's fabric side)."""
import csv
import glob
import json
import sys


def per_launch(dirname, counter, needle):
    f = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


fdir, wdir, needle, out = sys.argv[1:5]
fetch_kib, n1 = per_launch(fdir, "FETCH_SIZE", needle)
write_kib, n2 = per_launch(wdir, "WRITE_SIZE", needle)
res = {
    "kernel_substring": needle,
    "launches_sampled": [n1, n2],
    "fetch_bytes_per_launch": 2 * 1024 * fetch_kib,  # x2: gfx950 FETCH_SIZE under-count of wide coalesced reads
    "write_bytes_per_launch": 1024 * write_kib,
    "traffic_bytes_per_launch": 2 * 1024 * fetch_kib + 1024 * write_kib,
    "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; KiB units; FETCH_SIZE doubled (gfx950)",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
