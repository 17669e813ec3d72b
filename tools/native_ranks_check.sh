#!/bin/bash
# crt_render with N rank processes on ONE GPU (--host-exchange --same-device) against the single-process run, any mode:
#   tools/native_ranks_check.sh [ranks] [mode] [WxH]    -> "identical" or the first differing frame
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-3}; MODE=${2:-200}; SIZE=${3:-333x217}
OUT=$(mktemp -d)
EXE=$ROOT/directx-raytracer_amd/crt_render
SCENE=$ROOT/tests/golden/dragon.crtscene
ARGS="--mode $MODE --size $SIZE --frames 3 --orbit 7 --spp 3 --bounces 2"
$EXE $SCENE $ARGS --out $OUT/solo > $OUT/solo.log 2>&1 || { echo "single-process run failed"; cat $OUT/solo.log; exit 1; }
$EXE $SCENE $ARGS --out $OUT/ranks --ranks $N --host-exchange --same-device > $OUT/ranks.log 2>&1 || { echo "$N-rank run failed"; cat $OUT/ranks.log; exit 1; }
for f in 0 1 2; do cmp -s $OUT/solo_$f.ppm $OUT/ranks_$f.ppm || { echo "frame $f differs"; exit 1; }; done
echo "identical ($N ranks, mode $MODE, $SIZE)"
rm -rf $OUT
