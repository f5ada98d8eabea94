#!/bin/bash
# Same-box A/B of whole library builds on the la_gemm shapes: tools/lib_ab.sh <out dir> <lib a> <lib b> [...]; the processes alternate
# (a b c a b c ...), the table holds the best median of each build per shape.
OUT=$1; shift
mkdir -p $OUT
for i in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    LA_TOOLS_LIB=$PWD/$lib VARIANTS=${VARIANTS:-1} ROUNDS=${ROUNDS:-9} python tools/gemm_ab.py 2>/dev/null > $OUT/$n.$i.log
  done
done
python - $OUT "$@" <<'PY'
import glob, os, re, sys
out, libs = sys.argv[1], [os.path.basename(l)[:-3] for l in sys.argv[2:]]
best = {}
for n in libs:
    for f in glob.glob(f"{out}/{n}.*.log"):
        for line in open(f):
            m = re.match(r"(\S+)\s+(\S+):\s+v\d+\s+([\d.]+) us", line)
            if m:
                k = (m.group(1), m.group(2))
                best.setdefault(k, {}).setdefault(n, []).append(float(m.group(3)))
print("shape".ljust(28) + "".join(n.rjust(22) for n in libs))
for k, d in best.items():
    print((k[0] + " " + k[1]).ljust(28) + "".join(f"{min(d.get(n, [0])):10.1f} ({sorted(d.get(n, [0]))[len(d.get(n, [0])) // 2]:7.1f})".rjust(22) for n in libs))
PY
