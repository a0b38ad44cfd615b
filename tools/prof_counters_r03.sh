# SQ counters (MFMA pipe busy, waits, LDS) of the round-3 tile kernels and of the register-resident solver: separate --pmc passes,
# kernel trace only (MI355X_MICROARCH.md).  Outputs gpurun_out/r03_*_counters.csv
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16"
B="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
A64="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"
for W in simnn fmap; do
  CMD="python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --pmc $A -d gpurun_out/r03_pc_${W}_a -o s --output-format csv -- $CMD > gpurun_out/r03_pc_${W}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc $B -d gpurun_out/r03_pc_${W}_b -o s --output-format csv -- $CMD > gpurun_out/r03_pc_${W}_b.log 2>&1
done
rocprofv3 --kernel-trace --pmc $A64 -d gpurun_out/r03_pc_fmap_c -o s --output-format csv -- python bench.py --workload fmap --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03_pc_fmap_c.log 2>&1
python tools/pmc_counters.py gpurun_out/r03_simnn_counters.csv "simnn_pipe_kernel<64, 4, 0" gpurun_out/r03_pc_simnn_a/s_counter_collection.csv gpurun_out/r03_pc_simnn_b/s_counter_collection.csv
python tools/pmc_counters.py gpurun_out/r03_fmap_simnn4_counters.csv "simnn_pipe_kernel<64, 2, 3" gpurun_out/r03_pc_fmap_a/s_counter_collection.csv gpurun_out/r03_pc_fmap_b/s_counter_collection.csv
python tools/pmc_counters.py gpurun_out/r03_fmap_solver_counters.csv "fmap_solve_reg_kernel" gpurun_out/r03_pc_fmap_c/s_counter_collection.csv gpurun_out/r03_pc_fmap_b/s_counter_collection.csv
rm -rf gpurun_out/r03_pc_*_[abc]
cat gpurun_out/r03_simnn_counters.csv gpurun_out/r03_fmap_simnn4_counters.csv gpurun_out/r03_fmap_solver_counters.csv
