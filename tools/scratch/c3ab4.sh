for cfg in "64 68 68 128 256" "64 34 34 256 512" "64 136 136 64 128" "64 17 17 512 1024" "64 136 136 128 256" "16 68 68 128 256" "16 17 17 512 1024" "8 136 136 128 256"; do
  for v in 0 2; do echo -n "tall=$v  "; OM_C3_TALL=$v python tools/conv16_bench.py $cfg 3 1 res; done
done
