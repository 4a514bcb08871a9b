ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/prof_r06q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 11 --warmup 11 --no-cpu-baseline --no-kernel-pass --no-stream --no-side --no-fwd > $OUT/log.txt 2>&1
python $ROOT/scripts/trace_queue_sequence.py $OUT/trace 2 400 > $OUT/queue_sequence.txt 2>&1
rm -rf $OUT/trace
head -12 $OUT/queue_sequence.txt
