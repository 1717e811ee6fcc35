mkdir -p gpurun_out/r4d
for v in base fwdw4; do echo "== $v"; RFA_LIB_PATH=build/variants/$v/librfa_hip.so python tools/shape_sweep.py 8,1024,32,8,128,1 4,2048,32,8,128,1 1,2048,16,8,128,1 1,2048,2,1,128,1 1,4096,32,8,128,1 1,8192,32,8,128,1 2>&1 | grep "^| "; done | tee gpurun_out/r4d/fwdw4.txt
