#!/bin/bash
# Development aid: builds an A/B copy of the library whose render_impl<0> (the uninstrumented, no-CSG kernels)
# is compiled with extra flags, e.g.
#   tools/build_variant.sh stamps -DFRAY_STAMPS        ->  build/ab/stamps/libfrayhip.so
# and is selected at run time with FRAYHIP_LIB=build/ab/stamps/libfrayhip.so.  The other objects are the
# standard ones (run `make` first).  build/ is not in history but travels to the GPU box.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/ab/$NAME
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-machine-licm \
  -I$ROOT/include -I$ROOT/fray_amd/csrc -DFRAY_ST=0 "$@" -Rpass-analysis=kernel-resource-usage \
  -c $ROOT/fray_amd/csrc/render_variant.hip -o $OUT/variant0.o 2> $OUT/variant0.resources.txt || { cat $OUT/variant0.resources.txt; exit 1; }
# capi.hip sees DStats too: rebuild it when the flags change its layout
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-machine-licm \
  -I$ROOT/include -I$ROOT/fray_amd/csrc "$@" -c $ROOT/fray_amd/csrc/capi.hip -o $OUT/capi.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libfrayhip.so $OUT/capi.o $OUT/variant0.o $ROOT/fray_amd/csrc/capi_comm.o -ldl \
  $ROOT/fray_amd/csrc/variant1.o $ROOT/fray_amd/csrc/variant2.o $ROOT/fray_amd/csrc/variant3.o $ROOT/fray_amd/csrc/variant4.o $ROOT/fray_amd/csrc/variant5.o $ROOT/fray_amd/csrc/variant8.o $ROOT/fray_amd/csrc/variant9.o \
  $ROOT/fray_amd/csrc/host_scene.o $ROOT/fray_amd/csrc/host_loaders.o $ROOT/fray_amd/csrc/host_exr.o $ROOT/fray_amd/csrc/capi_host.o
python3 $ROOT/tools/kernel_resources.py $OUT/variant0.resources.txt > $OUT/resources.txt
echo "built $OUT/libfrayhip.so"
