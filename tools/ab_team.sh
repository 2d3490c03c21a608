set -e
cd /root/repo
timeout -k 10 300 python tools/team_probe.py 1,2,3,4,6 16 1250x1000x1,400x1000x1,40x2000x1 > gpurun_out/team_probe2.log 2>&1 || { tail -20 gpurun_out/team_probe2.log; exit 1; }
cat gpurun_out/team_probe2.log
timeout -k 10 200 python tools/dense_fuzz.py 120 3 > gpurun_out/dense_fuzz_team.log 2>&1 || { tail -20 gpurun_out/dense_fuzz_team.log; exit 1; }
tail -1 gpurun_out/dense_fuzz_team.log
