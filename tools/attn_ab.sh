#!/bin/bash
# tools/attn_ab.sh <out dir> <lib a> <lib b> ...: the builds alternate three times; best time per shape and build
OUT=$1; shift
mkdir -p $OUT
for i in 1 2 3; do for lib in "$@"; do LA_TOOLS_LIB=$PWD/$lib python tools/attn_ab.py 2>/dev/null > $OUT/$(basename $lib .so).$i.log; done; done
python - $OUT "$@" <<'PY'
import glob, os, re, sys
out, libs = sys.argv[1], [os.path.basename(l)[:-3] for l in sys.argv[2:]]
best, chk = {}, {}
for n in libs:
    for f in sorted(glob.glob(f"{out}/{n}.*.log")):
        for line in open(f):
            m = re.match(r"(.{28}) +([\d.]+) us .* checksum (\S+)", line)
            if m:
                best.setdefault(m.group(1).strip(), {}).setdefault(n, []).append(float(m.group(2)))
                chk.setdefault(m.group(1).strip(), {})[n] = m.group(3)
print("shape".ljust(30) + "".join(n.rjust(24) for n in libs))
for k, d in best.items():
    print(k.ljust(30) + "".join(f"{min(d.get(n, [0])):9.1f} {chk[k].get(n, '')}".rjust(24) for n in libs))
PY
