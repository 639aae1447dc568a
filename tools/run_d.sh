out=gpurun_out/r06_d; mkdir -p $out; export TMPDIR=/tmp REGTR_DEV=1
for v in h0r0 h1r0 h0r1 ""; do
  export REGTR_VARIANT=$v
  python tools/preprocess_bench.py --pairs 192 --reps 5 2>&1 | tail -1
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python tools/preprocess_bench.py --pairs 192 --reps 5 > $out/prof_$v.log 2>&1
  db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_${v:-default}.md 2>&1; rm -rf $out/prof
  grep -E "radius|k_insert|scatter_cells" $out/kernel_stats_${v:-default}.md
done
