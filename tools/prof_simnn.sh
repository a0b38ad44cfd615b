cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CMD="python bench.py --workload simnn --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_prof_simnn -o s --output-format csv -- $CMD > gpurun_out/r02_prof_simnn.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -d gpurun_out/r02_pmc_simnn_a -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_simnn_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/r02_pmc_simnn_b -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_simnn_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r02_pmc_simnn_c -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_simnn_c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02_pmc_simnn_f -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_simnn_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02_pmc_simnn_w -o s --output-format csv -- $CMD > gpurun_out/r02_pmc_simnn_w.log 2>&1
ls gpurun_out/r02_pmc_simnn_a gpurun_out/r02_prof_simnn
tail -2 gpurun_out/r02_pmc_simnn_a.log
