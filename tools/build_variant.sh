#!/bin/bash
# A second build of the library with one source compiled under other -D flags, for A/Bs of compile-time constants:
#   bash tools/build_variant.sh NAME xr_mlp.hip "-DBX_WAVES=16 -DBX_PF_OP=0"   ->  xrnerf_amd/libxrnerf_mi355_NAME.so
# then XRNERF_LIB=$PWD/xrnerf_amd/libxrnerf_mi355_NAME.so python bench.py ...   (the .so travels to the GPU box)
set -e
R=$(cd $(dirname $0)/.. && pwd); NAME=$1; SRC=$2; FLAGS=$3
python -m xrnerf_amd.build > /dev/null
EXTRA=$(python - <<PY
import sys; sys.path.insert(0, '$R')
from xrnerf_amd import build as b
print(' '.join(b.COMMON + b.SOURCES['$SRC']))
PY
)
O=$R/xrnerf_amd/build/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc $EXTRA $FLAGS -c $R/xrnerf_amd/csrc/$SRC -o $O
OBJS=""
for s in $(python - <<PY
import sys; sys.path.insert(0, '$R')
from xrnerf_amd import build as b
print(' '.join(b.SOURCES))
PY
); do if [ "$s" = "$SRC" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $R/xrnerf_amd/build/${s%.hip}.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/xrnerf_amd/libxrnerf_mi355_$NAME.so $OBJS
echo $R/xrnerf_amd/libxrnerf_mi355_$NAME.so
