mkdir -p gpurun_out/r4q
SH="16,512,32,8,128,1 8,1024,32,8,128,1 4,2048,32,8,128,1 2,4096,32,8,128,1 8,1024,32,32,128,1 4,2048,32,32,128,1"
for e in "X=0" "RFA_DKDV_WIDE=0" "RFA_DKDV_NSPLIT=1" "RFA_DKDV_WIDE=1 RFA_DKDV_NSPLIT=2"; do echo "== $e"; env $e python tools/shape_sweep.py $SH 2>&1 | grep "^| [0-9]"; done | tee gpurun_out/r4q/dkdv_plans_short.txt
