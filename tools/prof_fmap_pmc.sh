# SQ / cache counters of the four-map tile kernel inside the config-2 step (separate --pmc passes, no other tracing)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CMD="python bench.py --workload fmap --steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -d gpurun_out/r02_pmc_fmap_a -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_fmap_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d gpurun_out/r02_pmc_fmap_b -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_fmap_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r02_pmc_fmap_c -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_fmap_c.log 2>&1
rm -f gpurun_out/r02_pmc_fmap_[abc]/s_kernel_trace.csv
ls gpurun_out/r02_pmc_fmap_a
