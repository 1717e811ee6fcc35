# round 6, session 2: where the time of short launches goes (per-kernel trace of B8 S1024 / B4 S2048 / headline; fixed costs)
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 200 python tools/epilogue_cost.py > $O/epilogue_cost.txt 2>&1; echo "epilogue rc $?"
for spec in 8,1024,32,8,128,1 4,2048,32,8,128,1 1,8192,32,8,128,1 1,8192,32,8,128,0; do
  tag=$(echo $spec | tr , _)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o run -- python tools/shape_sweep.py $spec > $O/prof_$tag.log 2>&1
  db=$(find $O/prof_$tag -name '*.db' | head -1)
  python profiles/summarize_rocpd.py $db > $O/prof_$tag.txt 2>&1
  rm -rf $O/prof_$tag
done
cat $O/epilogue_cost.txt; for f in $O/prof_*.txt; do echo "== $f"; head -12 $f; done
