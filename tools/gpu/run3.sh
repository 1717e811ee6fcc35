mkdir -p gpurun_out/r4c
for lim in 2684354560 4831838208 9663676416; do echo "== limit $lim"; RFA_DS_SPILL_MAX_BYTES=$lim python tools/shape_sweep.py 1,16384,32,8,128,1 1,32768,32,8,128,1 1,32768,32,32,128,1 2>&1 | grep "^| 1"; done | tee gpurun_out/r4c/limit_sweep.txt
