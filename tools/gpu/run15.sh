mkdir -p gpurun_out/r4o
SH="8,1024,32,8,128,1 4,2048,32,8,128,1 2,4096,32,8,128,1 8,1024,32,32,128,1 16,512,32,8,128,1 8,1024,32,8,64,1 1,8192,32,8,128,1"
for f in auto 4x32; do echo "== RFA_FWD_FORM=$f"; RFA_FWD_FORM=$f python tools/shape_sweep.py $SH 2>&1 | grep "^| [0-9]"; done | tee gpurun_out/r4o/fwd_forms_short.txt
