#!/bin/bash
# Build A/B variants of librt_hip.so with extra kernel flags:  profiles/ab_build.sh <name> "<KFLAGS>" [product]
# -> build/ab/librt_hip_<name>.so   (bench with RT_HIP_LIB=<that path>); a test build (-DRT_TESTING) unless "product" is given
set -e
#    "hybrid": product kernels (96 VGPRs) under a test-build host layer, so that the host's environment switches exist (the probe fields
#    the test build appends to rt_launch are its LAST members: the product kernels read the prefix they know)
NAME=$1; FLAGS=$2; TESTING=-DRT_TESTING; APITESTING=
[ "${3:-}" = product ] && TESTING=
[ "${3:-}" = hybrid ] && { TESTING=; APITESTING=-DRT_TESTING; }
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/html5-canvas-raytracer_amd/csrc
OUT=$REPO/build/ab; mkdir -p $OUT
T=$(mktemp -d)
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -mllvm -disable-machine-licm $TESTING -I$SRC"
/opt/rocm/bin/hipcc $COMMON $FLAGS -DRT_STRICT=0 -ffp-contract=fast -c $SRC/rt_kernel.hip -o $T/kf.o &
/opt/rocm/bin/hipcc $COMMON $FLAGS -DRT_STRICT=1 -ffp-contract=off -c $SRC/rt_kernel.hip -o $T/ks.o &
/opt/rocm/bin/hipcc $COMMON $APITESTING $FLAGS -c $SRC/rt_api.hip -o $T/api.o &
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -Wall -I$SRC $FLAGS -x c++ -c $SRC/rt_tables.cpp -o $T/tables.o &
/opt/rocm/bin/hipcc $COMMON $FLAGS -ffp-contract=off -c $SRC/rt_tables_gpu.hip -o $T/tables_gpu.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/librt_hip_$NAME.so $T/api.o $T/tables.o $T/tables_gpu.o $T/kf.o $T/ks.o -ldl
cp $T/kf.o $OUT/rt_kernel_fast_$NAME.o      # for profiles/kernel_resources.sh / isa_histogram.sh
rm -rf $T
echo built $OUT/librt_hip_$NAME.so
