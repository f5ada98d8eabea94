cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/abw
rm -f $R/gpurun_out/abw/stats.log; python $R/tools/attn_bwd_bench.py > $R/gpurun_out/abw/wall.log 2>&1
for i in 0 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abw$i -o s -- python $R/tools/attn_bwd_bench.py $i > /tmp/abw$i.log 2>&1 || tail -5 /tmp/abw$i.log
  f=$(find /tmp/abw$i -name '*kernel_stats.csv' | head -1)
  echo "== shape $i" >> $R/gpurun_out/abw/stats.log
  python - "$f" >> $R/gpurun_out/abw/stats.log <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']: print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
done
cat $R/gpurun_out/abw/wall.log $R/gpurun_out/abw/stats.log
