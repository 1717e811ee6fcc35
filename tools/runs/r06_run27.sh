mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for spec in 8,1024,32,8,128,1 12,1024,32,8,128,1 6,1280,32,8,128,1; do
  tag=$(echo $spec | tr , _)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/pk_$tag -o run -- python tools/shape_sweep.py $spec > $O/pk_$tag.log 2>&1
  python profiles/summarize_rocpd.py $(find $O/pk_$tag -name '*.db' | head -1) 2>&1 | grep -E "fwd|persist" | cut -c1-110
  grep "^| [0-9]" $O/pk_$tag.log
  rm -rf $O/pk_$tag
done
