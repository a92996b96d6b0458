"""Instruction mix of a kernel in a hipcc -S listing, cut at its software barriers (the ds_add_u32 arrivals):
    python tools/isa_phases.py /tmp/mlp3.s k_mlp_step3
One row per segment in program order: VALU / transcendental / SALU / waitcnt / MFMA / LDS reads / LDS writes / global."""
import collections
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
cut = sys.argv[3] if len(sys.argv) > 3 else "ds_add_u32"
import re
m = re.search(r"^\S*" + re.escape(name) + r"\S*:", s, re.M)      # the (mangled) label that contains the name
i = m.start()
body = s[i:s.index("s_endpgm", i)]
lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";", "//"))]
segs, cur = [], collections.Counter()
for l in lines:
    op = l.split()[0]
    if op.endswith(":"):
        continue
    if op.startswith(cut):
        segs.append(cur)
        cur = collections.Counter()
    if op.startswith("v_mfma"):
        cur["mfma32" if "32x32" in op else "mfma16"] += 1
    elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        cur["trans"] += 1
    elif op.startswith("v_cvt_pk_bf16"):
        cur["cvt_pk"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
    elif op.startswith("s_waitcnt"):
        cur["wait"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith("ds_read") or op.startswith("ds_bpermute"):
        cur["lds_rd"] += 1
    elif op.startswith("ds_"):
        cur["lds_wr"] += 1
    elif op.startswith("scratch_"):
        cur["scratch"] += 1
    elif op.startswith(("global_", "buffer_")):
        cur["glob"] += 1
segs.append(cur)
keys = ["valu", "cvt_pk", "trans", "salu", "wait", "mfma32", "mfma16", "lds_rd", "lds_wr", "glob", "scratch"]
print("seg " + " ".join(f"{k:>7s}" for k in keys))
for n, c in enumerate(segs):
    print(f"{n:3d} " + " ".join(f"{c[k]:7d}" for k in keys))
