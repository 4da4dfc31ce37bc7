#!/bin/bash
# Round-6 measurement set in ONE gpurun call:  gpurun -- 'bash tools/profile_round6.sh r06_h'
# As tools/profile_round5.sh (per BASELINE config and the standing workloads: rocprofv3 --kernel-trace --stats of the CONTRACT loop,
# six PMC passes of the same loop, profiles/pmc_latest.json restamped with them, then the bench line that carries them), plus:
# the sparse formulation with the reference's OSQP leg as its CPU baseline, and the per-stage instruction
# census (tools/census.sh; needs variants/stop*/libqmpc.so built from THIS source by tools/census_build.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, pmc key, batch, bench args...
  local name=$1 key=$2 batch=$3; shift 3
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$name -o s --output-format csv -- \
      python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-extras "$@" > $OUT/stats_$name.log 2>&1
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  bash $R/tools/pmc.sh $TAG/pmc_$name --no-pipelined "$@" > $OUT/pmc_$name.log 2>&1
  cp $OUT/pmc_$name/pmc_summary.json $OUT/pmc_summary_$name.json 2>/dev/null
  rm -rf $OUT/stats_$name $OUT/pmc_$name/p[0-9]*
  (cd $R && python tools/pmc_to_latest.py $OUT/pmc_summary_$name.json $key $batch profiles/${TAG}_pmc_summary_$name.json $OUT/kernel_stats_$name.csv) >> $OUT/pmc_to_latest.log 2>&1
  python $R/bench.py --steps 300 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
}
run cfg1 config1 1024
run cfg2 config2 4096 --config 2
run trot_h16 config3 4096 --config 3
run cfg4 config4 8192 --config 4
run standing_h10 standing_h10 1024 --workload standing --horizon 10
run standing_h14 standing_h14 1024 --workload standing --horizon 14
for w in "large_trot_h36 long-trot 36 30" "large_stand_h36 long-stand 36 20" "large_standing_h24 standing 24 20"; do
  set -- $w
  bash $R/tools/pmc.sh $TAG/pmc_$1 --no-pipelined --workload $2 --horizon $3 --steps 5 --warmup 1 > $OUT/pmc_$1.log 2>&1
  cp $OUT/pmc_$1/pmc_summary.json $OUT/pmc_summary_$1.json 2>/dev/null
  rm -rf $OUT/pmc_$1/p[0-9]*
  (cd $R && python tools/pmc_to_latest.py $OUT/pmc_summary_$1.json ${2}_h$3 1024 profiles/${TAG}_pmc_summary_$1.json) >> $OUT/pmc_to_latest.log 2>&1
  python $R/bench.py --steps $4 --warmup 3 --workload $2 --horizon $3 --no-cpu-all-cores --no-pipelined > $OUT/bench_$1.json 2> $OUT/bench_$1.err
done
python $R/bench.py --steps 100 --workload long-trot --horizon 24 --no-cpu-all-cores > $OUT/bench_long_trot_h24.json 2> $OUT/bench_long_trot_h24.err
python $R/bench.py --steps 50 --workload long-bound --horizon 36 --no-cpu-all-cores > $OUT/bench_long_bound_h36.json 2> $OUT/bench_long_bound_h36.err
python $R/bench.py --steps 200 --config 0 --no-cpu-baseline > $OUT/bench_cfg0.json 2>/dev/null
for b in 4096 16384; do python $R/bench.py --steps 100 --batch $b --no-cpu-baseline --no-pipelined --no-closed-loop > $OUT/bench_cfg1_b$b.json 2>/dev/null; done
# the reference's SPARSE formulation with its OSQP leg (and the dense qpOASES pipeline) timed beside it
python $R/bench.py --steps 200 --model sparse > $OUT/bench_sparse_cfg1.json 2> $OUT/bench_sparse_cfg1.err
python $R/bench.py --steps 100 --model sparse --config 2 --no-closed-loop > $OUT/bench_sparse_cfg2.json 2> $OUT/bench_sparse_cfg2.err
python $R/tools/shim_latency.py > $OUT/shim_latency.json 2> $OUT/shim_latency.err
# per-stage shader-clock stamps (one stamped call each): one-round launch, the five-per-CU instantiation, the 96-row class
(python $R/tools/gpu_phases.py 1 1024; python $R/tools/gpu_phases.py 1 64; python $R/tools/gpu_phases.py 1 8192; python $R/tools/gpu_phases.py 2 4096; python $R/tools/gpu_phases.py 3 4096) > $OUT/phases.txt 2>&1
cp $R/profiles/pmc_latest.json $OUT/pmc_latest.json
if [ -f $R/variants/stop4/libqmpc.so ]; then bash $R/tools/census.sh $TAG/census > $OUT/census.log 2>&1; cp $OUT/census/census_table.md $OUT/census_table.md 2>/dev/null; fi
ls -la $OUT
