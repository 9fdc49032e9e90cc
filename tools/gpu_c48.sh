#!/bin/bash
# conv48r_kernel / conv48rp_kernel: correctness (16-bit kernel tests + networks), output hashes, overlapped forwards, benches
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_bsrn.py tests/test_gpu_big.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -6
python tools/dbg/batch_eq.py - 2>&1 | grep -E "equal|DIFF"
for mc in "team04_rlfn bf16" "team04_rlfn f16" "team18_bsrn f16"; do python tools/dbg/out_hash.py $mc 2>/dev/null | tail -1; done
python tools/dbg/streams_race.py team04_rlfn bf16 100 2>&1 | grep -E "mismatching"
python tools/dbg/streams_race.py team18_bsrn f16 100 2>&1 | grep -E "mismatching"
for m in "team18_bsrn f16 --tile 270x480" "team04_rlfn bf16" "team04_rlfn bf16 --sizes div2k --streams 1" "team18_bsrn f16 --sizes div2k --streams 1" "team04_rlfn bf16 --sizes div2k"; do
  set -- $m; python bench.py --model $1 --compute $2 $3 $4 $5 $6 --no-cpu-baseline --steps 20 > /tmp/b.json 2>/dev/null; python tools/show_bench.py /tmp/b.json | head -5 | cut -c1-170; done
