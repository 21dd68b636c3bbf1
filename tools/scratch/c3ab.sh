for cfg in "32 136 136 128 256" "32 68 68 128 256" "32 34 34 256 512" "32 17 17 512 1024" "32 136 136 64 128" "32 68 68 256 512" "32 34 34 512 1024"; do
  for v in 0 2; do echo -n "tall=$v  "; OM_C3_TALL=$v python tools/conv16_bench.py $cfg 3 1; done
done
