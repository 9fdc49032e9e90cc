#!/bin/bash
TAG=r03c; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
source <(sed -n '/^run_cfg() {/,/^}/p;/^pmc_cfg() {/,/^}/p' tools/profile_r03.sh)
cp profiles/pmc_traffic.json $O/pmc_traffic.json
run_cfg c4_bsrn_f16_270x480 --model team18_bsrn --compute f16 --tile 270x480
pmc_cfg c4_bsrn_f16_270x480 --model team18_bsrn --compute f16 --tile 270x480
