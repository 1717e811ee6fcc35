O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_rccl_world1.py -x -q -m gpu > $O/pytest_golden_rccl.log 2>&1; echo "pytest golden/rccl rc $?"; tail -3 $O/pytest_golden_rccl.log
timeout 300 python tools/plan_sweep.py > $O/plan_sweep.md 2>&1; echo "plan sweep rc $?"; tail -12 $O/plan_sweep.md
timeout 200 python tools/small_launch.py --rank 7 > $O/small_launch_rank7.txt 2>&1
timeout 200 python tools/small_launch.py --rank 3 > $O/small_launch_rank3.txt 2>&1
cat $O/small_launch_rank7.txt $O/small_launch_rank3.txt | grep -v amdgpu.ids
timeout 300 python tools/shape_sweep.py > $O/shape_sweep.md 2>&1; cat $O/shape_sweep.md | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/$O/prof_small7 -o t -- python $R/tools/small_launch.py --rank 7 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_d256 -o t -- python $R/tools/shape_sweep.py 1,8192,16,4,256,1 1,8192,20,5,192,1 8,1024,32,8,128,1 > /dev/null 2>&1
cd $R
for d in prof_small7 prof_d256; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; head -14 "$f" | cut -c1-200; done
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +20M -delete
