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
# the Makefile's flags; W14D=1 in the environment adds the dual-role kernel (conv_wino14d.hip)
FILES="conv_igemm conv_igemm_f16 conv3x3_f16 conv_wino conv_wino24 conv_wino14 conv_igemm_split conv_stem conv_stem2 preprocess coco_format post"
MODEL=build/om_model.o
if [ "$W14D" == "1" ]; then
  FILES="$FILES conv_wino14d"; FLAGS="$FLAGS -DOM_WITH_W14D=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DOM_WITH_W14D=1 -x hip -c om_model.cpp -o $B/om_model.o
  MODEL=$B/om_model.o
  if [ $# -gt 0 ]; then set -- "$@" conv_wino14 conv_wino14d; fi
fi
for F in $FILES; do
  EX=""; case "$F" in post|preprocess|coco_format) EX="-ffp-contract=off -fno-slp-vectorize";; conv_stem|conv_wino|conv_wino24) EX="-fno-slp-vectorize";; esac
  if [ -n "$NOSLP_ALL" ]; then case "$F" in post|preprocess|coco_format) ;; *) EX="-fno-slp-vectorize";; esac; fi     # round 6's A/B: the flag on every file
  FL=""
  if [ $# -eq 0 ] || [[ " $* " == *" $F "* ]]; then FL="$FLAGS"; fi
  if [ -z "$FL" ] && [ -f build/$F.o ]; then OBJS="$OBJS build/$F.o"; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EX $FL -c $F.hip -o $B/$F.o &
  OBJS="$OBJS $B/$F.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/$NAME.so $OBJS $MODEL
echo "ab/$NAME.so"
