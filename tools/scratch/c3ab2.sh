for cfg in "32 34 34 2048 1024" "32 68 68 1024 512" "8 136 136 1024 256"; do
  for v in 0 2; do echo -n "tall=$v  "; OM_C3_TALL=$v python tools/conv16_bench.py $cfg 3 1; done
done
