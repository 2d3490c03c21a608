set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph_build" 2>&1 | tail -3
for i in 1 2; do python3 tools/phase_probe.py --contigs 5000 --reps 5 2>&1 | tail -n 1 | cut -c1-420; done
python3 tools/phase_probe.py --contigs 625 --reps 5 2>&1 | tail -n 1 | cut -c1-420
