#!/bin/bash
# Side builds of the library with pieces of the fp32 stem kernel removed (timing experiments; results
# are numerically meaningless): tools/bin/libcilrs_hip_stemdbg<mask>.so, used through CILRS_LIB=.
# mask bits: 1 no LDS operand reads, 2 no row copies into LDS, 4 no stores of the result.
set -e
cd "$(dirname "$0")/../cilrs-autonomous-driving-carla_amd/csrc"
make -s
mkdir -p ../../tools/bin
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable \
      -DCILRS_STEM_DBG=$m -c stem_f32.hip -o /tmp/stem_f32_dbg$m.o
  objs=$(ls *.o | grep -v '^stem_f32.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcilrs_hip_stemdbg$m.so /tmp/stem_f32_dbg$m.o $objs
  echo built tools/bin/libcilrs_hip_stemdbg$m.so
done
