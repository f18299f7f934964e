#!/bin/bash
# Tuning build of the library (tune_int reads the environment): gpurun_ab/libsmvs_tune.so, loaded with SMVS_LIB_PATH.
cd "$(dirname "$0")/.." && mkdir -p gpurun_ab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -fvisibility=hidden -munsafe-fp-atomics -Wno-pass-failed -DSMVS_TUNING "$@" \
  $(for f in costvol costvol_bwd warp regress red costreg featnet filter; do echo satmvs_amd/csrc/$f.hip; done) -o gpurun_ab/libsmvs_tune.so
