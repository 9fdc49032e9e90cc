#!/bin/bash
# round 6, first GPU call: the recorder-based graph capture + dispatch-mode guard, the copy-roof family, the default bench line
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_multi.py tests/test_abi_and_host.py -q -x -m gpu 2>&1 | tail -15 > $O/graph_tests.txt
cat $O/graph_tests.txt
ESR_BW_PROBE_VERBOSE=1 timeout 300 python - > $O/copy_roof.txt 2>&1 <<'PY'
import ctypes, torch
from ntire2022_esr_amd import _lib as L
lib = L.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mb, reps in ((16, 64), (64, 16), (256, 4), (1024, 1), (2048, 1)):
    buf = torch.empty(2 * (mb << 20), dtype=torch.uint8, device="cuda")
    g = ctypes.c_double(0)
    for _ in range(2):
        L.check(lib.esr_bw_probe(ctypes.c_void_p(buf.data_ptr()), mb << 20, reps, st, ctypes.byref(g)), "bw")
    print(f"== 2 x {mb} MiB x {reps}: best {g.value:.1f} GB/s", flush=True)
    del buf
PY
grep "==" $O/copy_roof.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
