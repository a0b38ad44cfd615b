# Round-2 measurement pass on the GPU box: bench lines for every workload, rocprofv3 kernel stats and the HBM counter
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/r02_*.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for W in fmap simnn zoomout stress icp; do
  python bench.py --workload $W > gpurun_out/r02_bench_$W.log 2>&1
  tail -1 gpurun_out/r02_bench_$W.log | cut -c1-400
done
python bench.py --gpus 2 --single-device --steps 5 --warmup 2 > gpurun_out/r02_bench_2rank.log 2>&1
for W in fmap simnn zoomout stress icp; do
  S="--steps 6 --warmup 2"; [ $W = zoomout ] && S="--steps 1 --warmup 1"; [ $W = stress ] && S="--steps 3 --warmup 1"
  CMD="python bench.py --workload $W $S --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --stats -d gpurun_out/r02_prof_$W -o s --output-format csv -- $CMD > gpurun_out/r02_prof_$W.log 2>&1
  if [ $W = fmap ] || [ $W = simnn ] || [ $W = stress ]; then
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02_pmc_${W}_f -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_${W}_f.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02_pmc_${W}_w -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_${W}_w.log 2>&1
  fi
  # the traces are large: only the summaries travel back
  rm -f gpurun_out/r02_prof_$W/s_kernel_trace.csv gpurun_out/r02_pmc_${W}_f/s_kernel_trace.csv gpurun_out/r02_pmc_${W}_w/s_kernel_trace.csv
done
ls gpurun_out | grep r02_ | head -40
