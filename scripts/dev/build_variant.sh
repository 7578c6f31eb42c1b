#!/bin/bash
# scripts/dev/build_variant.sh NAME "-DFLAG=.. ..." : the library with other compile-time constants -> isocon_amd/lib/libisocon_hip_NAME.so
# (git-ignored, travels with gpurun); run with ISOCON_LIB=isocon_amd/lib/libisocon_hip_NAME.so python bench.py ...
R=$(cd "$(dirname "$0")/../.." && pwd)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function $2 -o $R/isocon_amd/lib/libisocon_hip_$1.so $R/isocon_amd/csrc/isocon_hip.hip
