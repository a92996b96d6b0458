"""Fold the rocprofv3 passes of tools/pmc_mfma.sh into one JSON object per kernel (names containing <kernel substr>, split by
grid size):  parse_mfma_pmc.py <kernel substr> <expected MFMA pipe cycles per launch | 0> <out.json> <dir>...

Counters are means over the kernel's dispatches.  Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES
counts shader cycles, summed over every SIMD of the chip (= 32 x N for N v_mfma_f32_32x32x16_bf16, 16 x N for 16x16x32);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of their
busy cycles, so GRBM_GUI_ACTIVE / 8 = the dispatch's length in shader cycles and / 8 / duration = the clock it ran at.
    mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)      share of the dispatch the matrix pipes worked
    of_bf16_peak   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)       the same against the 2.5 PFLOP/s the peak is quoted at
`expected`: what the kernel's instruction counts say the counter should read (bench.py's roofline.pipe uses the same count)."""
import collections
import csv
import glob
import json
import re
import sys

kern, expect, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
kerns = kern.split("|")            # several substrings: "a|b"
N_SIMD, PEAK_HZ = 1024, 2.4e9
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)


def key_of(name, grid):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
    return (m.group(1) if m else name.split("(")[0][:110]) + f" grid={grid}"


for d in sys.argv[4:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(kq in r["Kernel_Name"] for kq in kerns):
                vals[key_of(r["Kernel_Name"], r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        if not d.rstrip("/").endswith("a"):
            continue                       # durations from the pass that carries the MFMA counter
        for r in csv.DictReader(open(f)):
            if any(kq in r["Kernel_Name"] for kq in kerns):
                durs[key_of(r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append(
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
res = {}
for k, v in vals.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    o = {"dispatches": {n: len(x) for n, x in v.items()}, "counters_mean": {n: round(x, 1) for n, x in c.items()}}
    dk = durs.get(k) or next((dv for dn, dv in durs.items() if dn.split(" grid=")[0] == k.split(" grid=")[0]), [])
    dur = sum(dk) / len(dk) if dk else None
    o["duration_us_under_the_profiler"] = round(dur * 1e6, 2) if dur else None
    busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
    if busy is not None and gui:
        o["mfma_pipe_busy"] = round(busy / (N_SIMD * gui / 8.0), 4)
        if dur:
            o["clock_GHz"] = round(gui / 8.0 / dur / 1e9, 3)
            o["of_bf16_peak"] = round(busy / (N_SIMD * dur * PEAK_HZ), 4)
    if expect > 0 and busy is not None:
        o["expected_mfma_busy_cycles"] = expect
        o["counter_over_expected"] = round(busy / expect, 4)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in c:
                o[n.lower() + "_share_of_wave_cycles"] = round(c[n] / wc, 4)
    res[k] = o
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
