#!/bin/bash
# SQ / LDS counters of the three dominant kernels, one counter group per rocprofv3 pass (no other trace domains).
#   gpurun --timeout 1500 -- 'bash tools/pmc_summary.sh r02'   ->  gpurun_out/<round>_pmc_summary.txt
set -u
R=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$R
mkdir -p $OUT
SUM=gpurun_out/${R}_pmc_summary.txt
: > $SUM
# (round 6: the folded-LayerNorm forms at the model's 96-image shapes and the product's attention kernels: PMC_SHAPES="qkv_n lin1_n proj_p lin2_p attn_rows attn_win")
for w in ${PMC_SHAPES:-lin1 lin2 qk v2 attn}; do
  i=0
  for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/$w -o p$i -- python tools/gemm_probe.py $w > /dev/null 2>&1
  done
  python - "$w" $OUT/$w >> $SUM <<'PY'
import csv, glob, sys
from collections import defaultdict
w, d = sys.argv[1], sys.argv[2]
tot, cnt = defaultdict(float), defaultdict(int)
kern = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if not ("gemm_" in name or "attn_fwd" in name):
            continue
        kern = name.split("(")[0]
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[row["Counter_Name"]] += 1
print(f"== {w}: {kern}  (per-launch means over {max(cnt.values()) if cnt else 0} launches)")
for k in sorted(tot):
    print(f"  {k:28s} {tot[k] / cnt[k]:.4g}")
m = {k: tot[k] / cnt[k] for k in tot}
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (32 cycles per 32x32x16 MFMA)
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"  -> kernel duration {cyc:.4g} shader cycles; matrix pipe busy = MFMA_BUSY / (cycles x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")
if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
    print(f"  -> LDS bank-conflict cycles / LDS active cycles = {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.4f}")
if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
    print(f"  -> wave time waiting on any instruction     = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
if "FETCH_SIZE" in m:
    print(f"  -> HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = {(2 * m['FETCH_SIZE'] + m.get('WRITE_SIZE', 0)) / 1024:.0f} MiB")
PY
done
cat $SUM
