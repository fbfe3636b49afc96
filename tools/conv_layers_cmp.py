"""Layer-by-layer comparison of two `tools/conv_layers.py` outputs: `python tools/conv_layers_cmp.py A.txt B.txt`."""
import re, sys
def parse(f):
    out=[]
    for l in open(f):
        m=re.match(r"n_out=\s*(\d+)\s+(\d+)->\s*(\d+) pairs=\s*(\d+)\s+([\d.]+) us\s+useful\s+([\d.]+) TF",l)
        if m: out.append((int(m[1]),int(m[2]),int(m[3]),int(m[4]),float(m[5]),float(m[6])))
    return out
o=parse(sys.argv[1]); n=parse(sys.argv[2])
for a,c in zip(o,n):
    print(f"{a[0]:7d} {a[1]:4d}->{a[2]:4d} pairs {a[3]:8d} A {a[4]:7.1f}us {a[5]:6.1f}TF   B {c[4]:7.1f}us {c[5]:6.1f}TF  x{a[4]/c[4]:.2f}")
print("total A", sum(a[4] for a in o), "B", sum(a[4] for a in n))
