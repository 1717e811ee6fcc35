for v in base av1 av2 av4 av6 base; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v"; timeout 300 python tools/fwd_persist_check.py 1,8192,32,8,1 1,8192,32,8,0 4,2048,32,8,1 2>&1 | grep "^| [0-9]" | head -3
done
