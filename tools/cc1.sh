#!/bin/bash
# compile ONE translation unit of the library to /tmp with its gfx950 assembly kept (iteration helper): tools/cc1.sh esr_chain
set -e
cd "$(dirname "$0")/../ntire2022_esr_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . -c "$1.hip" -o "/tmp/$1.o" -save-temps=obj 2>&1 | grep -v "^$" | grep -v "warning generated" || true
S="/tmp/$1-hip-amdgcn-amd-amdhsa-gfx950.s"
[ "$S" -nt "$1.hip" ] || { echo "cc1.sh: $1.hip did NOT compile (stale or missing $S)"; exit 1; }
grep -n "\.vgpr_count\|\.agpr_count\|\.private_segment_fixed_size\|    \.name:" "$S" | paste - - - - | sed 's/ \+/ /g' | head -40
