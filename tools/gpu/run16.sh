mkdir -p gpurun_out/r4p
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p/kt_short -o kt -- python $R/tools/shape_sweep.py 8,1024,32,8,128,1 > $R/gpurun_out/r4p/kt_short.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p/kt_llama -o kt -- python $R/tools/small_launch.py > $R/gpurun_out/r4p/kt_llama.log 2>&1
cd $R
for n in kt_short kt_llama; do echo "== $n"; python profiles/summarize_rocpd.py $(find gpurun_out/r4p/$n -name "*_results.db" | head -1) | cut -c1-150; done | tee gpurun_out/r4p/summary.txt
find gpurun_out/r4p -name "*.db" -delete
