mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
SH="16,512,32,8,128,1 8,1024,32,8,128,1 4,2048,32,8,128,1 2,4096,32,8,128,1 4,4096,32,8,128,1 8,2048,32,8,128,1 2,8192,32,8,128,1 1,8192,32,8,128,1 2,4096,16,16,128,1"
rm -f $O/batch_order.txt
for v in base batchslow base batchslow; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v" >> $O/batch_order.txt
  timeout 300 python tools/shape_sweep.py $SH >> $O/batch_order.txt 2>&1
done
unset RFA_LIB_PATH
cat $O/batch_order.txt
timeout 600 python tools/bal_check.py 4,4096,32,8 8,2048,32,8 12,1024,32,8 2,8192,32,8 8,1024,32,8 4,2048,32,8 > $O/bal_check3.txt 2>&1; cat $O/bal_check3.txt
