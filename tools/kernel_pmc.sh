#!/bin/bash
# SQ / LDS counters of the kernels whose name contains <substring>, one counter group per rocprofv3 pass (counters only: no other trace
# domains).   tools/kernel_pmc.sh <tag> <kernel name substring> <command ...>   ->  gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; SUB=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
SUM=gpurun_out/pmc_$TAG.txt
rm -rf $OUT; mkdir -p $OUT
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT -o p$i -- "$@" > /dev/null 2>&1
done
python - "$SUB" $OUT > $SUM <<'PY'
import csv, glob, sys
from collections import defaultdict
sub, d = sys.argv[1], sys.argv[2]
tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if sub not in name:
            continue
        k = name.split("(")[0][:110]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in tot:
    m = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    print(f"== {k}  (per-launch means over {max(cnt[k].values())} launches)")
    for c in sorted(m):
        print(f"  {c:28s} {m[c]:.4g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        print(f"  -> kernel duration {cyc:.4g} shader cycles; matrix pipe busy = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM"):
            if c in m:
                print(f"  -> {c} / SQ_WAVE_CYCLES = {m[c] / m['SQ_WAVE_CYCLES']:.3f}")
PY
cat $SUM
