for v in base pin0 base pin0; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v"; timeout 200 python tools/shape_sweep.py 1,8192,20,5,192,1 1,8192,16,4,160,1 1,8192,20,5,192,0 2>/dev/null | grep "^| [0-9]"
done
unset RFA_LIB_PATH
timeout 200 python tools/shape_sweep.py 1,8192,16,4,256,1 2,4096,8,8,256,1 2>/dev/null | grep "^| [0-9]"
timeout 600 python - <<'PY'
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'ring-flash-attention_amd')
import plan_sweep
plan_sweep.fwd_points=lambda q: []
pts=[p for p in plan_sweep.bwd_points(False) if len(p)>7]
plan_sweep.bwd_points=lambda q: pts
plan_sweep.sweep(False)
PY
timeout 900 python -m pytest tests/test_gpu_head_dim_256.py -x -q -m gpu 2>&1 | tail -3
