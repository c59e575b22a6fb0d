R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hilam128 -- python3 $R/bench.py --model hi_lam --hidden-dim 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_hilam128.out 2> $R/gpurun_out/prof_hilam128.err
find $R/gpurun_out/prof_hilam128 -name "*kernel_trace.csv" -delete
