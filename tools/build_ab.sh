#!/bin/bash
# Development aid: a complete A/B copy of the library built from a tree (default: this working tree; or a git revision) with extra
# HIP flags, selected at run time with FRAYHIP_LIB=build/ab/NAME/libfrayhip.so.  Uses the Makefile's own flags (EXTRA_HIPFLAGS), so
# an A/B build cannot drift from the shipped one.  build/ is not in history but travels to the GPU box.
#   tools/build_ab.sh NAME [-r REV] [extra hipcc flags...]
set -e
NAME=$1; shift
REV=""
if [ "$1" = "-r" ]; then REV=$2; shift 2; fi
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=/tmp/fray_ab_$NAME
rm -rf $WORK; mkdir -p $WORK
if [ -n "$REV" ]; then
  git -C $ROOT archive $REV Makefile include fray_amd/csrc | tar -x -C $WORK
else
  mkdir -p $WORK/fray_amd/csrc $WORK/include
  cp $ROOT/Makefile $WORK/; cp $ROOT/include/*.h $WORK/include/
  cp $ROOT/fray_amd/csrc/*.hpp $ROOT/fray_amd/csrc/*.h $ROOT/fray_amd/csrc/*.hip $ROOT/fray_amd/csrc/*.cpp $WORK/fray_amd/csrc/
fi
make -s -C $WORK -j8 fray_amd/libfrayhip.so EXTRA_HIPFLAGS="$*" > $WORK/build.log 2>&1 || { tail -30 $WORK/build.log; exit 1; }
OUT=$ROOT/build/ab/$NAME
mkdir -p $OUT
cp $WORK/fray_amd/libfrayhip.so $OUT/
python3 $ROOT/tools/kernel_resources.py $WORK/fray_amd/csrc/variant*.resources.txt > $OUT/resources.txt
echo "built $OUT/libfrayhip.so ($*)"
