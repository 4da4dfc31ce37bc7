cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/size_order_study.py > $R/gpurun_out/size_order_study3.json 2> $R/gpurun_out/size_order_study3.err
for o in plain size_desc fit_desc_then_pass_desc pass_last interleaved_desc; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/so_$o -o t --output-format csv -- python $R/tools/size_order_study.py cfg4 $o > $R/gpurun_out/so_$o.log 2>&1
  f=$(find $R/gpurun_out/so_$o -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/so_${o}_kernel_stats.csv; rm -rf $R/gpurun_out/so_$o
done
