#!/bin/bash
# An alternative fmm.hip (kept outside csrc/ while it is being worked on) linked with the product's other objects:
#   tools/build_alt.sh <alt fmm.hip> <out .so> [extra hipcc flags]      then  DAZIM_LIB=<out .so> python ...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
python -c "import dazimsurftomo_amd as dz; dz.build()" > /dev/null
obj=/tmp/alt_fmm_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$root/dazimsurftomo_amd/csrc $3 -c -o $obj $1
others=$(ls $root/dazimsurftomo_amd/lib/obj/libdazim_hip.*.o | grep -v "fmm.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $2 $obj $others -lrccl
rm -f $obj
echo built $2
