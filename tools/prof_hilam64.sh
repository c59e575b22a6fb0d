R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hilam64 -- python3 $R/bench.py --model hi_lam --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_hilam64.out 2> $R/gpurun_out/prof_hilam64.err
find $R/gpurun_out/prof_hilam64 -name "*kernel_trace.csv" -delete
