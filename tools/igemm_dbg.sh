#!/bin/bash
# Side builds of the library with pieces of the implicit-GEMM K loop removed (timing experiments;
# results are numerically meaningless): tools/bin/libcilrs_hip_dbg<mask>.so (use with CILRS_LIB=).
# mask bits: 1 no global loads, 2 no LDS stores, 4 no barriers, 8 no LDS reads, 16 ADD a fused BatchNorm-apply to the A operand.
set -e
cd "$(dirname "$0")/../cilrs-autonomous-driving-carla_amd/csrc"
make -s
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable \
      -DCILRS_IGEMM_DBG=$m -c conv_igemm.hip -o /tmp/conv_igemm_dbg$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcilrs_hip_dbg$m.so \
      /tmp/conv_igemm_dbg$m.o conv_wgrad.o bn_pool.o heads_optim.o heads_gemm.o augment.o infer_f16.o wgrad_f16.o stem_f16.o conv_small.o net.o
  echo built tools/bin/libcilrs_hip_dbg$m.so
done
