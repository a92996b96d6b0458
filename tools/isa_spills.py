"""Where a kernel spills: scratch loads / stores bucketed by how many MFMAs precede them.  python tools/isa_spills.py file.s kernel-substring"""
import collections
import sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index(":", i)
body = s[i:s.index("s_endpgm", i)]
lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";", "//"))]
idx = [k for k, l in enumerate(lines) if l.startswith("v_mfma")]
print("instructions", len(lines), "first / last mfma at", idx[0], idx[-1], "mfma count", len(idx))
h = collections.Counter()
for k, l in enumerate(lines):
    if l.startswith("scratch_"):
        n = sum(1 for x in idx if x < k)
        h[(n // 12) * 12, l.split()[0][:13]] += 1
for k in sorted(h):
    print(k, h[k])
c = collections.Counter()
for l in lines[idx[0]:idx[-1]]:
    op = l.split()[0]
    c["mfma" if op.startswith("v_mfma") else op if op.startswith(("ds_", "scratch_", "global_", "v_accvgpr")) else "valu" if op.startswith("v_") else "s_waitcnt" if op.startswith("s_waitcnt") else "salu"] += 1
print(dict(c.most_common(30)))
