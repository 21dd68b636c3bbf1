#!/bin/bash
# Build a variant of the library for same-box A/B runs (tools/ab_bench.sh):  tools/build_variant.sh NAME "-DOM_EXP_..." [file ...]
# -> ab/NAME.so (ab/ is git-ignored; it travels to the GPU box with gpurun).  The named source files (default: all) get the flags.
set -e
NAME=$1; FLAGS="$2 -DOM_MEASUREMENT_BUILD=1"; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/ab/build_$NAME
mkdir -p $B
cd $R/orienmask_amd/csrc
OBJS=""
for F in conv_igemm conv_igemm_f16 conv3x3_f16 conv_wino conv_wino24 conv_wino14 conv_wino14d conv_igemm_split conv_stem conv_stem2 preprocess coco_format post; do
  EX=""; case "$F" in post|preprocess|coco_format) EX="-ffp-contract=off -fno-slp-vectorize";; conv_stem|conv_wino|conv_wino24) EX="-fno-slp-vectorize";; esac
  FL=""
  if [ $# -eq 0 ] || [[ " $* " == *" $F "* ]]; then FL="$FLAGS"; fi
  if [ -z "$FL" ] && [ -f build/$F.o ]; then OBJS="$OBJS build/$F.o"; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EX $FL -c $F.hip -o $B/$F.o &
  OBJS="$OBJS $B/$F.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/$NAME.so $OBJS build/om_model.o
echo "ab/$NAME.so"
