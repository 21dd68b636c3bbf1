for cfg in "32 68 68 1024 512" "32 136 136 128 256"; do
  for v in 0 2; do 
    echo -n "tall=$v rand  "; OM_C3_TALL=$v python tools/conv16_bench.py $cfg 3 1
    echo -n "tall=$v zero  "; ZERO_X=1 OM_C3_TALL=$v python tools/conv16_bench.py $cfg 3 1
  done
done
