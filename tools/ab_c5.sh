set -e
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/ab_tests.log 2>&1 || { tail -20 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
timeout -k 10 200 python tools/dense_fuzz.py > gpurun_out/ab_fuzz.log 2>&1 || { tail -20 gpurun_out/ab_fuzz.log; exit 1; }
tail -2 gpurun_out/ab_fuzz.log
for i in 1 2; do
AASM_LIB_OVERRIDE=/root/repo/variants/base.so timeout -k 10 120 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'], d['value'])"
timeout -k 10 120 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['value'])"
done
