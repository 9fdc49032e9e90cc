#!/bin/bash
# A research build of the library for A/B runs: tools/r06/build_variant.sh <name> <unit> "<extra hipcc flags>"  ->  tools/r06/libesr_<name>.so
# (<unit>.hip is recompiled with the flags, every other object is the product's from build/obj; load it with ESR_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../.."
name=$1; unit=$2; flags=$3
mkdir -p build/obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I ntire2022_esr_amd/csrc $flags -c ntire2022_esr_amd/csrc/$unit.hip -o build/obj_$name/$unit.o 2>&1 | grep -v "warning generated\|^$" || true
objs=""
for o in build/obj/esr_*.o; do b=$(basename $o); if [ "$b" = "$unit.o" ]; then objs="$objs build/obj_$name/$unit.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/r06/libesr_$name.so $objs
echo built tools/r06/libesr_$name.so
