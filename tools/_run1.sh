cd /root/repo
cp labelanything_amd/libla_hip.so /tmp/d9.so
for v in d9 d8 base; do
  [ $v = d9 ] && cp /tmp/d9.so labelanything_amd/libla_hip.so || cp labelanything_amd/libla_gelu_$v.so labelanything_amd/libla_hip.so
  echo "===== GELU $v"
  timeout 600 python tools/parity_report.py 2>&1 | grep "precise=auto=patch+vmean+projmean+neck\] " | grep "dec=float32" | cut -c1-330
  timeout 900 python tests/test_parity_seeds_gpu.py 11 12 13 14 15 2>&1 | grep -v amdgpu.ids | tail -12
done
cp /tmp/d9.so labelanything_amd/libla_hip.so
for i in 1 2; do
echo "== d9"; timeout 300 python tools/normfold_ab.py 2>&1 | grep -i "lin1" | head -3
echo "== base"; LA_TOOLS_LIB=$PWD/labelanything_amd/libla_gelu_base.so timeout 300 python tools/normfold_ab.py 2>&1 | grep -i "lin1" | head -3
done
