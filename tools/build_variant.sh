#!/bin/bash
# Development: a second library with match.hip compiled under extra flags, for same-box A/B runs through DIF_LIB
# (tools/match_exp.sh: the -DB1_EXP=N timing experiments of match_b1_kernel).  Needs an up-to-date build/ (build.py).
#   tools/build_variant.sh e2 -DB1_EXP=2     ->  deep-insight-face_amd/lib/libdif_e2.so
# (remove the variants before a gpurun call that does not need them: every .so travels with the snapshot)
set -e
cd "$(dirname "$0")/../deep-insight-face_amd"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c csrc/match.hip -o build/match_$name.o
objs=$(ls build/*.hip.o | grep -v match.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libdif_$name.so $objs build/match_$name.o
echo built lib/libdif_$name.so
