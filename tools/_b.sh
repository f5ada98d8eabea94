mkdir -p gpurun_out/r02b
for w in cfg3_train cfg5 cfg4 cfg3 cfg1; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02b/bench_$w.json 2> gpurun_out/r02b/bench_$w.err
  echo "== $w rc=$?"; cut -c1-330 gpurun_out/r02b/bench_$w.json; tail -3 gpurun_out/r02b/bench_$w.err | grep -v amdgpu
done
