set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph_build" 2>&1 | tail -15 > gpurun_out/tl/parity_gb.txt || { cat gpurun_out/tl/parity_gb.txt; exit 1; }
cat gpurun_out/tl/parity_gb.txt
python3 tools/kprof_gb.py 5000 1000 21 2>&1 | tail -2
python3 tools/phase_probe.py --contigs 5000 --reps 5 2>&1 | tail -n 1 | cut -c1-700
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
