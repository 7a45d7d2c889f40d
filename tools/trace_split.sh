cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for P in f32 bf16-mixed; do
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$P -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prithvi --no-bf16 --no-profile --precision $P > $R/gpurun_out/trace_$P.log 2>&1
python3 $R/tools/stream_split.py $(find $R/gpurun_out/trace_$P -name "*kernel_trace.csv" | head -1) --steps 8 --tail-us ${TAIL_US:-0} > $R/gpurun_out/split_$P.txt 2>&1
rm -rf $R/gpurun_out/trace_$P
cat $R/gpurun_out/split_$P.txt
done
