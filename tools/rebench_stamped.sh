R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${1:-rebench}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift; python $R/bench.py --steps 300 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run cfg1
run standing_h10 --workload standing --horizon 10
run standing_h14 --workload standing --horizon 14
run standing_h16 --workload standing --horizon 16
run trot_h16 --config 3
run cfg2 --config 2
run cfg4 --config 4
ls -la $OUT
