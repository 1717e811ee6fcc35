for v in base dqs8 dqs5 dqsnt0 base dqs8 dqs5 dqsnt0; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v"; timeout 100 python tools/small_launch.py --rank 7 2>/dev/null | grep "^| 8 "; timeout 100 python tools/small_launch.py --rank 3 2>/dev/null | grep "^| 8 "
done
