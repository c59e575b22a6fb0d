R=$GRAFT_REPO_ROOT
python $R/tools/micro_m2m.py > $R/gpurun_out/micro_m2m.json 2> $R/gpurun_out/micro_m2m.err; echo micro_exit=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/micro_m2m.py --iters 3 > /dev/null 2> $R/gpurun_out/pmc1.err; echo pmc1_exit=$?
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/micro_m2m.py --iters 3 > /dev/null 2> $R/gpurun_out/pmc2.err; echo pmc2_exit=$?
ls $R/gpurun_out/pmc1/*/ | head
