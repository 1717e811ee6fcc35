mkdir -p gpurun_out/r06b
timeout 600 python tools/fwd_persist_check.py > gpurun_out/r06b/fwd_persist.txt 2>&1; echo rc $?; grep -v amdgpu.ids gpurun_out/r06b/fwd_persist.txt | tail -24
