# Round-4 bench pass on the GPU box: the driver's default line, every other workload, the fp32-basis and two-rank variants and the
# per-step kernel tables.  Outputs under gpurun_out/r04b/ (copied to profiles/ by hand).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b; O=gpurun_out/r04b
python bench.py > $O/r04_fmap_bench.json 2> $O/fmap.err
for W in simnn zoomout stress icp surface_map; do python bench.py --workload $W --no-secondary > $O/r04_${W}_bench.json 2> $O/$W.err; done
python bench.py --basis f32 --no-secondary --no-cpu-baseline > $O/r04_fmap_f32basis_bench.json 2> $O/f32.err
python bench.py --gpus 2 --single-device --no-secondary --no-cpu-baseline > $O/r04_fmap_2rank_single_device_bench.json 2> $O/2rank.err
for W in zoomout fmap icp stress; do python tools/step_profile.py $W > $O/r04_${W}_step_kernels.txt 2> /dev/null; done
python tools/simnn_power_check.py > $O/r04_simnn_power_test.txt 2> /dev/null
python tools/proj_check.py > $O/r04_proj_onepass_test.txt 2> /dev/null
ls -la $O | tail -20
