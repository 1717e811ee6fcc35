mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for spec in 8,1024,32,8,128,1 4,2048,32,8,128,1 4,4096,32,8,128,1; do
  tag=$(echo $spec | tr , _)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/after_$tag -o run -- python tools/shape_sweep.py $spec > $O/after_$tag.log 2>&1
  echo "== $spec"; python profiles/summarize_rocpd.py $(find $O/after_$tag -name '*.db' | head -1) 2>&1 | head -8 | cut -c1-120
  grep "^| [0-9]" $O/after_$tag.log
  rm -rf $O/after_$tag
done
