#!/bin/bash
# Diagnostic build of the library (-DOX_DIAG): the wrong-result timing switches (OX_AF_DBG: skip the pair loop / the
# epilogue of the fused assemble_first) exist ONLY in this build, never in oasisx_amd/liboasisx_hip.so.
#   tools/build_diag.sh && OX_LIB_PATH=$PWD/tools/liboasisx_hip_diag.so OX_AF_DBG=1 python tools/af_bench.py
set -e
cd "$(dirname "$0")/../oasisx_amd/csrc"
out=../../tools/liboasisx_hip_diag.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fconstexpr-steps=50000000 -shared -DOX_DIAG -o "$out" \
  ox_spmv.hip ox_ksp.hip ox_assemble.hip ox_dist.hip ox_setup.hip -L"${ROCM_PATH:-/opt/rocm}/lib" -lrccl \
  -Wl,-rpath,"${ROCM_PATH:-/opt/rocm}/lib"
echo "built $out"
