#!/bin/bash
# Diagnostic build of the library (-DOX_DIAG): the wrong-result timing switches (OX_AF_DBG: skip the pair loop / the
# epilogue of the fused assemble_first) exist ONLY in this build, never in oasisx_amd/liboasisx_hip.so.
#   tools/build_diag.sh && OX_LIB_PATH=$PWD/tools/liboasisx_hip_diag.so OX_AF_DBG=1 python tools/af_bench.py
#   tools/build_diag.sh W   -> tools/liboasisx_hip_diagW.so with -DOX_DIAG_W as well: the pair loop of assemble_first
#                              reads a per-cell record of 42 doubles (what a cell-centric pre-pass G . u_ab would
#                              store) instead of geometry, cell dofs and coefficient gathers
set -e
cd "$(dirname "$0")/../oasisx_amd/csrc"
extra=""
out=../../tools/liboasisx_hip_diag.so
if [ "$1" = "W" ]; then extra="-DOX_DIAG_W"; out=../../tools/liboasisx_hip_diagW.so; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fconstexpr-steps=50000000 -shared -DOX_DIAG $extra -o "$out" \
  ox_spmv.hip ox_ksp.hip ox_assemble.hip ox_dist.hip ox_setup.hip -L"${ROCM_PATH:-/opt/rocm}/lib" -lrccl \
  -Wl,-rpath,"${ROCM_PATH:-/opt/rocm}/lib"
echo "built $out"
