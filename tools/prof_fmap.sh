cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=${1:-fmap}
CMD="python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_prof_$W -o s --output-format csv -- $CMD > gpurun_out/r02_prof_$W.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02_pmc_${W}_f -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_${W}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02_pmc_${W}_w -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_${W}_w.log 2>&1
cat gpurun_out/r02_prof_$W/s_kernel_stats.csv
