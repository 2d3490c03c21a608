set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph_build" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -3
for gl in 0 1; do python3 tools/phase_probe.py --contigs 5000 --reps 3 --heavy 1 --graph-launches $gl 2>&1 | tail -n 1 | cut -c1-420; done
python3 tools/phase_probe.py --contigs 5000 --reps 3 2>&1 | tail -n 1 | cut -c1-420
