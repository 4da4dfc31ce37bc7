#!/bin/bash
# One round's measurement set, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r02_a'
# For configs[1] and the standing / horizon-16 workloads (size classes 1, 2, 3, 4):
#   bench line (+ CPU baseline), rocprofv3 --kernel-trace --stats summary of the same command, PMC passes.
# Everything lands under gpurun_out/<tag>/ ; copy what should be judged into profiles/ (tools/collect_profiles.sh), then re-run the five
# bench lines with the restamped profiles/pmc_latest.json in place (tools/rebench_stamped.sh): the lines of THIS script carry no PMC fields.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  python $R/bench.py --steps 300 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$name -o s --output-format csv -- \
      python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-extras "$@" > $OUT/stats_$name.log 2>&1
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  bash $R/tools/pmc.sh $TAG/pmc_$name --no-pipelined "$@" > $OUT/pmc_$name.log 2>&1
  cp $OUT/pmc_$name/pmc_summary.json $OUT/pmc_summary_$name.json 2>/dev/null
  rm -rf $OUT/stats_$name $OUT/pmc_$name/p[0-9]*
}
run cfg1
run standing_h10 --workload standing --horizon 10
run standing_h14 --workload standing --horizon 14
run standing_h16 --workload standing --horizon 16
run trot_h16 --config 3
run cfg2 --config 2
run cfg4 --config 4
# long horizons (the 192-row class + the decoupled engine): trot at 24 segments, bounding-type gait at 36
python $R/bench.py --steps 100 --workload long-trot --horizon 24 --no-cpu-all-cores > $OUT/bench_long_trot_h24.json 2> $OUT/bench_long_trot_h24.err
python $R/bench.py --steps 50 --workload long-bound --horizon 36 --no-cpu-all-cores > $OUT/bench_long_bound_h36.json 2> $OUT/bench_long_bound_h36.err
# the large-problem path (192 < n_r <= 432): a trot at 36 segments, all feet down at 24 (braking) and 36 segments
python $R/bench.py --steps 30 --warmup 3 --workload long-trot --horizon 36 --no-cpu-all-cores --no-pipelined > $OUT/bench_large_trot_h36.json 2> $OUT/bench_large_trot_h36.err
python $R/bench.py --steps 20 --warmup 3 --workload standing --horizon 24 --no-cpu-all-cores --no-pipelined > $OUT/bench_large_standing_h24.json 2> $OUT/bench_large_standing_h24.err
python $R/bench.py --steps 20 --warmup 3 --workload long-stand --horizon 36 --no-cpu-all-cores --no-pipelined > $OUT/bench_large_stand_h36.json 2> $OUT/bench_large_stand_h36.err
for w in "large_trot_h36 long-trot 36 30" "large_stand_h36 long-stand 36 20"; do
  set -- $w
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$1 -o s --output-format csv -- \
      python $R/bench.py --steps $4 --warmup 3 --repeats 3 --workload $2 --horizon $3 --no-cpu-baseline --no-pipelined > $OUT/stats_$1.log 2>&1
  find $OUT/stats_$1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$1.csv \;
  rm -rf $OUT/stats_$1
  # (round 4: counters for the large-problem producer as well -- FETCH_SIZE / WRITE_SIZE behind the "memory-system limit" of DESIGN 3.6)
  bash $R/tools/pmc.sh $TAG/pmc_$1 --no-pipelined --workload $2 --horizon $3 --steps 5 --warmup 1 > $OUT/pmc_$1.log 2>&1
  cp $OUT/pmc_$1/pmc_summary.json $OUT/pmc_summary_$1.json 2>/dev/null
  rm -rf $OUT/pmc_$1/p[0-9]*
done
python $R/tools/jcqp_long.py 2>/dev/null | grep -v amdgpu > $OUT/jcqp_long.txt
python $R/tools/dense_threshold.py 2>/dev/null | grep -v amdgpu > $OUT/dense_threshold.txt
# the same standing workloads on the one-kernel path (before / after of the decoupled path in ONE profile set)
for hh in 10 14 16; do
  QMPC_NO_SPLIT=1 python $R/bench.py --steps 200 --workload standing --horizon $hh --no-cpu-baseline --no-pipelined > $OUT/bench_standing_h${hh}_one_kernel.json 2>/dev/null
done
python $R/tools/split_check.py 1024 2>/dev/null | grep -v "^{" | grep -v amdgpu > $OUT/split_check.txt
python $R/tools/engine_phase.py s10 1024 2>/dev/null | grep -v amdgpu > $OUT/engine_phases.txt
python $R/tools/engine_phase.py s14 1024 2>/dev/null | grep -v amdgpu >> $OUT/engine_phases.txt
# (the sweep kernels' stage stamps: robots the engine kernel did not stamp over)
QMPC_PHASE_SWEEP_ONLY=1 python $R/tools/gpu_phases.py s10 1024 2>/dev/null | grep -v amdgpu | head -7 > $OUT/sweep_phases.txt
QMPC_PHASE_SWEEP_ONLY=1 python $R/tools/gpu_phases.py s14 1024 2>/dev/null | grep -v amdgpu | head -7 >> $OUT/sweep_phases.txt
# bench lines only for the remaining BASELINE configs (one GPU's shard), batch scaling, calm standing, caller-side pipeline
for c in 0; do python $R/bench.py --steps 200 --config $c --no-cpu-baseline > $OUT/bench_cfg$c.json 2>/dev/null; done
for a in "--config 4 --batch 8192" "--workload standing --horizon 10 --batch 1024" "--workload standing --horizon 16 --batch 1024"; do python $R/tools/class_stats.py $a; done > $OUT/class_stats.txt 2>/dev/null
for b in 256 4096 16384 65536; do python $R/bench.py --steps 100 --batch $b --no-cpu-baseline --no-pipelined > $OUT/bench_cfg1_b$b.json 2>/dev/null; done
python $R/bench.py --steps 100 --caller-side fused --no-cpu-baseline > $OUT/bench_caller_fused.json 2>/dev/null
python $R/tools/shim_latency.py > $OUT/shim_latency.json 2> $OUT/shim_latency.err
python $R/tools/warm_rollout.py --cycles 30 > $OUT/warm_rollout.json 2> $OUT/warm_rollout.err
ls -la $OUT
