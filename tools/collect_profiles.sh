#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/<tag>/ (copied to profiles/ afterwards).
#   gpurun -- 'bash tools/collect_profiles.sh r05'   (needs alignasm_amd/libalignasm_amd_kprof.so: make -C alignasm_amd/csrc ../libalignasm_amd_kprof.so)
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $O/${TAG}_c3_bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/${TAG}_c3_bench_under_rocprof.json 2> $O/kt.err || exit 1
cp $O/kt/*/*kernel_stats.csv $O/${TAG}_c3_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_write.err || exit 1
python3 $R/tools/pmc_summary.py $O/${TAG}_c3_pmc_fetch_write.json $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_sq.err || exit 1
python3 $R/tools/pmc_summary.py $O/${TAG}_c3_pmc_sq.json $O/pmc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/kt5.err || exit 1
cp $O/kt5/*/*kernel_stats.csv $O/${TAG}_c5_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc5_fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc5_write.err || exit 1
python3 $R/tools/pmc_summary.py $O/${TAG}_c5_pmc_fetch_write.json $O/pmc5_fetch $O/pmc5_write
cp $O/${TAG}_c5_pmc_fetch_write.json $R/profiles/          # (bench.py reads roofline.traffic from profiles/)
python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --cpu-sample 200 > $O/${TAG}_c5_bench_file_1gpu.json 2> $O/c5.err || exit 1
for n in 1 250 625 1000 1250 2500; do python3 $R/tools/phase_probe.py --contigs $n --reps 3 2>&1 | tail -n 1 > $O/${TAG}_probe_n$n.json; done
for n in 1 625; do python3 $R/tools/phase_probe.py --contigs $n --reps 3 --chain-own-queue 1 2>&1 | tail -n 1 > $O/${TAG}_probe_n${n}_own_queue.json; done   # the chain class without the order wave (aasm_k67_chain3)
for n in 1 250 625 1000; do python3 $R/tools/phase_probe.py --contigs $n --reps 3 --chain none 2>&1 | tail -n 1 > $O/${TAG}_probe_n${n}_three_launches.json; done   # the chain class off: K6, pre-pass, K7 one after the other
python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 --graph-launches 1 2>&1 | tail -n 1 > $O/${TAG}_probe_c3_graph_by_separate_launches.json   # aasm_k46_graph off: row_fill, scan, rev_fill, rev_place, rev_hdr
python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 2>&1 | tail -n 1 > $O/${TAG}_probe_n5000.json
python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 --dup 3 --shuffle 1 2>&1 | tail -n 1 > $O/${TAG}_probe_c3_dup3_shuffled.json
python3 $R/tools/phase_probe.py --contigs 1000 --k 10000 --reps 3 2>&1 | tail -n 1 > $O/${TAG}_probe_k10000_1000contigs.json
python3 $R/tools/phase_probe.py --contigs 5000 --k 10000 --reps 3 2>&1 | tail -n 1 > $O/${TAG}_probe_k10000_5000contigs.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktk -- python3 $R/tools/phase_probe.py --contigs 5000 --k 10000 --reps 3 > /dev/null 2> $O/ktk.err || exit 1
cp $O/ktk/*/*kernel_stats.csv $O/${TAG}_c3_k10000_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktd -- python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 --dup 3 --shuffle 1 > /dev/null 2> $O/ktd.err || exit 1
cp $O/ktd/*/*kernel_stats.csv $O/${TAG}_c3_dup3_kernel_stats.csv
AASM_LIB_OVERRIDE=$R/alignasm_amd/libalignasm_amd_kprof.so python3 $R/tools/sortfix_probe.py > $O/${TAG}_sort_replay_sections.txt 2>&1
python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 --heavy 1 2>&1 | tail -n 1 > $O/${TAG}_probe_c3_heavy_tail.json
python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 --heavy 1 --chain none 2>&1 | tail -n 1 > $O/${TAG}_probe_c3_heavy_tail_three_launches.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kth -- python3 $R/tools/phase_probe.py --contigs 5000 --reps 3 --heavy 1 > /dev/null 2> $O/kth.err || exit 1
cp $O/kth/*/*kernel_stats.csv $O/${TAG}_c3_heavy_tail_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python3 $R/tools/phase_probe.py --contigs 625 --reps 5 > /dev/null 2> $O/kts.err || exit 1
cp $O/kts/*/*kernel_stats.csv $O/${TAG}_n625_kernel_stats.csv
python3 $R/tools/giant_probe.py > $O/${TAG}_giant_contigs.jsonl 2>&1
python3 $R/tools/e2e_cli.py --contigs 5000 --k 4 > $O/${TAG}_e2e_cli.log 2>&1
AASM_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/${TAG}_c4_selflaunch_2ranks_1gpu_gloo.json 2> $O/c4.err
python3 $R/tools/kprof.py 5000 1000 4 0 21 > $O/${TAG}_k7_k9_sections_c3.txt 2>&1
python3 $R/tools/kprof_gb.py 5000 1000 21 > $O/${TAG}_k46_graph_sections_c3.txt 2>&1
python3 $R/tools/kprof.py 5000 1000 4 0 21 3 > $O/${TAG}_k7_k9_sections_c3_dup3.txt 2>&1
python3 $R/tools/kprof_enum.py 5000 1000 10000 0 21 > $O/${TAG}_k8_sections.txt 2>&1
rm -rf $O/kt $O/kt5 $O/ktk $O/ktd $O/kth $O/kts $O/pmc5_fetch $O/pmc5_write $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
