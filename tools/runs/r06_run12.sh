mkdir -p gpurun_out/r06b
timeout 600 python tools/bal_check.py > gpurun_out/r06b/bal_check.txt 2>&1; echo "rc $?"; cat gpurun_out/r06b/bal_check.txt | tail -30
