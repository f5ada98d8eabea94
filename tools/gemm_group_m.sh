cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LA_TOOLS_LIB=$R/labelanything_amd/libla_hip_dbg.so
for g in 1 2 4 8 16; do
  echo "== LA_GEMM_GROUP_M=$g"
  LA_GEMM_GROUP_M=$g python $R/tools/gemm_group_m.py 2>/dev/null | grep x
  LA_GEMM_GROUP_M=$g rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/gm$g -o f -- python $R/tools/gemm_group_m.py > /dev/null 2>&1
  python - $(find /tmp/gm$g -name 'f_counter_collection.csv' | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_t256' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        k = r['Kernel_Name'][:40] + ' grid ' + r['Grid_Size']
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
for k, (n, v) in acc.items():
    print(f"   {k}: {n} launches, FETCH_SIZE {v / n / 1e6 * 2 / 1024:.2f} GB/launch after the x2 correction (raw KiB {v / n:.0f})")
PY
done
