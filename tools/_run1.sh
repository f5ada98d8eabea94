cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4
for i in 1 2; do
python bench.py --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('fused cs', d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if k in ('attn_fwd_rows','colmean16','gemm','colsum_fold')})"
python bench.py --no-cpu-baseline --no-eager-baseline --no-win-fused-cs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('colmean16', d['value'], d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if k in ('attn_fwd_rows','colmean16','gemm','colsum_fold')})"
done
