#!/bin/bash
# A measurement build of libbatrack_ba.so with one source compiled with extra flags (the other objects are the default build's):
#   tools/build_variant.sh <name> <source.hip> <extra flags...>   ->  _var/<name>/libbatrack_ba.so   (use with BT_LIB_PATH=_var/<name>/libbatrack_ba.so)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -c "import __graft_entry__ as g; g.build()"
mkdir -p _var/$name
obj=batrack_amd/lib/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -x hip -c batrack_amd/csrc/$src -o _var/$name/${src%.*}.o "$@"
objs=""
for o in $obj/*.o; do b=$(basename $o); if [ "$b" = "${src%.*}.o" ]; then objs="$objs _var/$name/$b"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o _var/$name/libbatrack_ba.so
echo "_var/$name/libbatrack_ba.so"
