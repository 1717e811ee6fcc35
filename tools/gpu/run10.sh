mkdir -p gpurun_out/r4j
python tools/small_launch.py 2>&1 | grep -v "^\[" | tee gpurun_out/r4j/small_launch.txt
python tools/small_launch.py --rank 0 2>&1 | grep "^|" | tee -a gpurun_out/r4j/small_launch.txt
RFA_FWD_FORM=8x32 python tools/small_launch.py 2>&1 | grep "^|" | tee gpurun_out/r4j/small_launch_8x32.txt
