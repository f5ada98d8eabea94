#!/bin/bash
# SQ / LDS counters of the four-wave GEMM (and its ablations) on one shape, one counter group per rocprofv3 pass.
#   gpurun --timeout 900 -- 'bash tools/w4_pmc.sh cube "2 0x1002 0x2002 1"'  ->  gpurun_out/w4_pmc_<shape>.txt
set -u
W=${1:-cube}
VARS=${2:-"2 1"}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_w4
SUM=gpurun_out/w4_pmc_$W.txt
: > $SUM
for v in $VARS; do
  rm -rf $OUT/$W; mkdir -p $OUT/$W
  i=0
  for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1))
    LA_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/$W -o p$i -- python tools/gemm_probe.py $W > /dev/null 2>&1
  done
  python - "$W" "$v" $OUT/$W >> $SUM <<'PY'
import csv, glob, sys
from collections import defaultdict
w, v, d = sys.argv[1], sys.argv[2], sys.argv[3]
tot, cnt = defaultdict(float), defaultdict(int)
kern = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if "gemm_" not in name:
            continue
        kern = name.split("(")[0]
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[row["Counter_Name"]] += 1
print(f"== {w} variant {v}: {kern}  (per-launch means over {max(cnt.values()) if cnt else 0} launches)")
m = {k: tot[k] / cnt[k] for k in tot}
for k in sorted(m):
    print(f"  {k:28s} {m[k]:.4g}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"  -> kernel duration {cyc:.4g} shader cycles; matrix pipe busy = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")
if "SQ_WAVE_CYCLES" in m:
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"):
        if k in m:
            print(f"  -> {k} / SQ_WAVE_CYCLES = {m[k] / m['SQ_WAVE_CYCLES']:.3f}")
PY
done
cat $SUM
