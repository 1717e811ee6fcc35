mkdir -p gpurun_out/r4n
for v in base alt alt4 a4 sfa3 base; do echo "== $v"; LD_LIBRARY_PATH=build/variants/$v timeout 300 ./tests/native/selftest 2>&1 | tail -1; RFA_LIB_PATH=build/variants/$v/librfa_hip.so python tools/shape_sweep.py 1,8192,32,8,128,1 1,8192,32,8,128,0 2>&1 | grep "^| 1"; done | tee gpurun_out/r4n/variants.txt
