#!/bin/bash
# Side builds of the library with pieces of the persistent bf16 convolution kernel removed (timing
# experiments; results are numerically meaningless): tools/bin/libcilrs_hip_c16dbg<mask>.so, used
# through CILRS_LIB=.  mask bits: 1 no multiplies / LDS fragment reads, 2 no operand loads,
# 4 no epilogue.
set -e
cd "$(dirname "$0")/../cilrs-autonomous-driving-carla_amd/csrc"
make -s
mkdir -p ../../tools/bin
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable \
      -DCILRS_CONV16_DBG=$m -c conv16.hip -o /tmp/conv16_dbg$m.o
  objs=$(ls *.o | grep -v '^conv16.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcilrs_hip_c16dbg$m.so /tmp/conv16_dbg$m.o $objs
  echo built tools/bin/libcilrs_hip_c16dbg$m.so
done
