#!/bin/bash
# research builds of the library with wino8_tail_f32_kernel's ablation switches (W8T_ABL bits, esr_wino.hip) -> tools/abl/libesr_w8t_<n>.so
#   bash tools/r06/build_w8t_variants.sh 0 1 2 ...        (after __graft_entry__.build(): the other objects come from build/obj)
cd "$(dirname "$0")/../.."; mkdir -p tools/abl build/w8t
OBJS=$(ls build/obj/*.o | grep -v esr_wino.o)
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I ntire2022_esr_amd/csrc $W8T_EXTRA -DW8T_ABL=$n -c ntire2022_esr_amd/csrc/esr_wino.hip -o build/w8t/esr_wino_$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/libesr_w8t_$n.so $OBJS build/w8t/esr_wino_$n.o && echo built $n ) &
done
wait
