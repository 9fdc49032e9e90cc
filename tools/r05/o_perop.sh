#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $O; cd $GRAFT_REPO_ROOT
for a in "4 bf16 32 256x256" "4 bf16 1 339x510" "0 bf16 32 256x256" "0 bf16 1 339x510" "18 f16 32 270x480" "18 f16 1 339x510" "-1 f32 32 256x256"; do
  timeout 200 python tools/per_op.py $a 2>&1 | grep -v amdgpu.ids >> $O/per_op.txt
done
