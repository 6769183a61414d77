# Pricing of a cell-centric pre-pass for assemble_first on the UNSTRUCTURED bench mesh (jittered 33^3 lattice, Delaunay,
# refined twice: 14.0 M tets, 18.9 M P2 rows), row-block launch.  Diagnostic builds only (wrong results, timing).
# The two diagW lines need the -DOX_DIAG_W code of commit b947ab9 (removed after the measurement: profiles/r06_assemble_first_delaunay_pricing.txt).
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/price_af_delaunay.txt
: > $OUT
D=$PWD/tools/liboasisx_hip_diag.so
W=$PWD/tools/liboasisx_hip_diagW.so
run() { echo "== $1" >> $OUT; shift; env "$@" MODES=1 python tools/af_bench.py 32 delaunay 2 2>&1 | grep -E "assemble_first|rows," >> $OUT; }
run "product library" OX_AF_DBG=0
run "diag build, dbg=0" OX_LIB_PATH=$D OX_AF_DBG=0
run "diag: no pair loop (dbg=1)" OX_LIB_PATH=$D OX_AF_DBG=1
run "diag: no epilogue (dbg=2)" OX_LIB_PATH=$D OX_AF_DBG=2
run "diag: coefficient gathers L1 hits (dbg=8)" OX_LIB_PATH=$D OX_AF_DBG=8
run "diag: + no cell-dof loads (dbg=24)" OX_LIB_PATH=$D OX_AF_DBG=24
run "diag: + geometry of 8 cells (dbg=28)" OX_LIB_PATH=$D OX_AF_DBG=28
run "diagW: pair loop reads the 42-double cell record (pre-pass emulation)" OX_LIB_PATH=$W OX_AF_DBG=0
run "diagW: the same, no epilogue" OX_LIB_PATH=$W OX_AF_DBG=2
cat $OUT
