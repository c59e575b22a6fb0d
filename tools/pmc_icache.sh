# instruction-cache behaviour of the fused kernels (gfx950: 64 KB I-cache per CU pair)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $R/gpurun_out/pmc_ic -- python3 $R/tools/micro_m2m.py --iters 3 > /dev/null 2> $R/gpurun_out/pmc_ic.err; echo pmc_ic_exit=$?
tail -3 $R/gpurun_out/pmc_ic.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_inst -- python3 $R/tools/micro_m2m.py --iters 3 > /dev/null 2> $R/gpurun_out/pmc_inst.err; echo pmc_inst_exit=$?
tail -3 $R/gpurun_out/pmc_inst.err
find $R/gpurun_out/pmc_ic $R/gpurun_out/pmc_inst -name "*kernel_trace.csv" -delete
