O=gpurun_out/r06
mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python tools/plan_sweep.py --dump $O/plan_sweep_dump.json > $O/plan_sweep_full.md 2>&1; echo "plan sweep rc $?"; tail -30 $O/plan_sweep_full.md
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof_small7 -o t -- python $R/tools/small_launch.py --rank 7 > $R/$O/prof_small7.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_small3 -o t -- python $R/tools/small_launch.py --rank 3 > $R/$O/prof_small3.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof_d256 -o t -- python $R/tools/shape_sweep.py 1,8192,16,4,256,1 1,8192,20,5,192,1 8,1024,32,8,128,1 > $R/$O/prof_d256.log 2>&1
cd $R
for d in prof_small7 prof_small3 prof_d256; do db=$(find $O/$d -name "*_results.db" | head -1); echo "== $d $db"; python profiles/summarize_rocpd.py $db > $O/$d.txt 2>&1; head -30 $O/$d.txt | cut -c1-220; done
rm -rf $O/prof_small7 $O/prof_small3 $O/prof_d256
