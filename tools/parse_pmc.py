"""Fold rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into per-launch HBM bytes for
one kernel, with the gfx950 corrections of MI355X_MICROARCH.md section HBM: counters are in KiB;
FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE
is exact for 16-B-per-lane streaming stores.  Usage: parse_pmc.py <kernel substr> <out.json> <dir>..."""
import csv, glob, json, sys
kern, out = sys.argv[1], sys.argv[2]
vals = {}
for d in sys.argv[3:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {"kernel": kern, "launches": {k: len(v) for k, v in vals.items()}}
fetch = sum(vals.get("FETCH_SIZE", [0])) / max(1, len(vals.get("FETCH_SIZE", [1]))) * 1024
write = sum(vals.get("WRITE_SIZE", [0])) / max(1, len(vals.get("WRITE_SIZE", [1]))) * 1024
res.update(fetch_size_raw_bytes=fetch, write_size_bytes=write, fetch_corrected_bytes=2 * fetch,
           hbm_bytes_per_launch=2 * fetch + write,
           note="FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), counters in KiB; separate --pmc passes")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
