"""Instruction mix of one kernel in a hipcc -S listing:  python tools/isa_mix.py /tmp/mlp3.s k_mlp_step3 [first_label last_label]"""
import collections
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name)
i = s.index(":", i)
body = s[i:s.index("s_endpgm", i)]
lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";", "//"))]
c = collections.Counter()
for l in lines:
    op = l.split()[0]
    if op.endswith(":"):
        continue
    key = op
    if op.startswith("v_mfma"):
        key = "mfma " + op
    elif op.startswith(("scratch_", "ds_", "global_", "buffer_")):
        key = op
    elif op.startswith("v_"):
        key = "valu"
    elif op.startswith("s_waitcnt"):
        key = "s_waitcnt"
    elif op.startswith("s_"):
        key = "salu"
    c[key] += 1
print(len(lines), "lines")
for k, v in c.most_common(40):
    print(f"{v:6d}  {k}")
