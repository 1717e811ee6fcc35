O=gpurun_out/r06
mkdir -p $O
for v in base nofuse base nofuse; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v"; timeout 200 python tools/shape_sweep.py 1,8192,16,4,256,1 1,8192,16,4,256,0 1,8192,20,5,192,1 2,4096,8,8,256,1 1,8192,16,4,160,1 2>/dev/null | grep "^| [0-9]"
done
unset RFA_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_head_dim_256.py -x -q -m gpu 2>&1 | tail -5
