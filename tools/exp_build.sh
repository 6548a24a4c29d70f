#!/bin/bash
# Side build of the library with the tuning switches of earlier rounds read from the environment
# again (-DCILRS_EXPERIMENTS; the product build compiles them to their defaults):
# tools/bin/libcilrs_hip_exp.so, used through CILRS_LIB=.  Objects go to /tmp so the product
# build's objects stay untouched.  Extra flags: exp_build.sh -DCILRS_WINO_ONE_PHASE=1 ...
set -e
cd "$(dirname "$0")/../cilrs-autonomous-driving-carla_amd/csrc"
OBJ=/tmp/cilrs_exp_obj; mkdir -p $OBJ ../../tools/bin
SRCS=$(sed -n 's/^SRCS := //p' Makefile)
for f in $SRCS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function \
      -DCILRS_EXPERIMENTS "$@" -c $f -o $OBJ/${f%.hip}.o &
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcilrs_hip_exp.so $OBJ/*.o
echo built tools/bin/libcilrs_hip_exp.so
