timeout 900 python -m pytest tests/test_gpu_plan_rules.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/fwd_persist_check.py 6,1280,32,8,1 8,1024,32,8,1 4,2048,32,8,1 5,1024,32,8,1 3,2048,32,8,1 12,1024,32,8,1 2>&1 | grep "^| [0-9]"
