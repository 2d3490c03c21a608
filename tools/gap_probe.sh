# Diagnostic: GPU idle gaps inside a C3 step (kernel trace of a short bench run, analysed by tools/gap_probe.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf $R/gpurun_out/gap && mkdir -p $R/gpurun_out/gap
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gap -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras ${1:+--contigs $1} > /dev/null 2> $R/gpurun_out/gap/err.txt || exit 1
python3 $R/tools/gap_probe.py $(ls -t $R/gpurun_out/gap/*/*kernel_trace.csv | head -1)
