#!/bin/bash
# rebuilds the library and the conv_trace harness together (they share the ConvArgs struct)
set -e
cd "$(dirname "$0")/../.."
python deep-insight-face_amd/build.py | tail -1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deep-insight-face_amd/csrc tools/ubench/conv_trace.hip \
  -L deep-insight-face_amd/lib -ldif -Wl,-rpath,'$ORIGIN/../../deep-insight-face_amd/lib' -o tools/ubench/conv_trace
