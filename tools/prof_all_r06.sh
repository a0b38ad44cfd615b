# Round-6 measurement pass on the GPU box: rocprofv3 kernel stats and the HBM counter passes (FETCH_SIZE / WRITE_SIZE in
# separate runs, as MI355X_MICROARCH.md prescribes) for the bench workloads.  Outputs under gpurun_out/r06_*; the traffic
# summaries carry the hash of the kernel sources they were measured on (bench.py quotes them only for that code).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for W in ${WORKLOADS:-fmap simnn zoomout stress}; do
  S="--steps 6 --warmup 2"; [ $W = zoomout ] && S="--steps 1 --warmup 1"; [ $W = stress ] && S="--steps 3 --warmup 1"
  CMD="python bench.py --workload $W $S --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_$W -o s --output-format csv -- $CMD > gpurun_out/r06_prof_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r06_pmc_${W}_f -o s --output-format csv -- $CMD > gpurun_out/r06_pmc_${W}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r06_pmc_${W}_w -o s --output-format csv -- $CMD > gpurun_out/r06_pmc_${W}_w.log 2>&1
  python tools/pmc_summary.py gpurun_out/r06_pmc_${W}_f/s_counter_collection.csv gpurun_out/r06_pmc_${W}_w/s_counter_collection.csv gpurun_out/r06_${W}_hbm_traffic_pmc.csv $CMD > /dev/null
  cp gpurun_out/r06_prof_$W/s_kernel_stats.csv gpurun_out/r06_${W}_kernel_stats.csv
  rm -rf gpurun_out/r06_prof_$W gpurun_out/r06_pmc_${W}_f gpurun_out/r06_pmc_${W}_w
done
ls gpurun_out | grep r06_ | head -40
# the documented call: one raw pair (bench.py --workload surface_map) and 64 raw pairs per call (tools/surface_map_streams.py)
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_sm1 -o s --output-format csv -- python bench.py --workload surface_map --no-cpu-baseline > gpurun_out/r06_prof_sm1.log 2>&1
cp gpurun_out/r06_prof_sm1/s_kernel_stats.csv gpurun_out/r06_surface_map_single_kernel_stats.csv; rm -rf gpurun_out/r06_prof_sm1
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_sm64 -o s --output-format csv -- python tools/surface_map_streams.py 1 > gpurun_out/r06_prof_sm64.log 2>&1
cp gpurun_out/r06_prof_sm64/s_kernel_stats.csv gpurun_out/r06_surface_map_batch64_kernel_stats.csv; rm -rf gpurun_out/r06_prof_sm64
ls gpurun_out | grep r06_ | head -40
