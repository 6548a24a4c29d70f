#!/bin/bash
# Side builds of the library with pieces of the Winograd kernel removed (timing experiments;
# results are numerically meaningless): tools/bin/libcilrs_hip_wdbg<mask>.so (use with CILRS_LIB=).
# mask bits: 1 no global loads, 2 no transform + LDS stores of a chunk, 4 no MFMAs, 8 no LDS fragment reads.
set -e
cd "$(dirname "$0")/../cilrs-autonomous-driving-carla_amd/csrc"
make -s
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable \
      -DCILRS_WINO_DBG=$m -c conv_wino.hip -o /tmp/conv_wino_dbg$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcilrs_hip_wdbg$m.so \
      /tmp/conv_wino_dbg$m.o conv_igemm.o conv_wgrad.o bn_pool.o heads_optim.o heads_gemm.o augment.o infer_f16.o wgrad_f16.o stem_f16.o conv_small.o infer_b1.o net.o
  echo built tools/bin/libcilrs_hip_wdbg$m.so
done
