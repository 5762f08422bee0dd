#!/bin/bash
# the two PMC passes of scripts/gpu_round.sh (HBM traffic per kernel), alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o r2 -- python $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_fetch.log 2>&1); echo "pmc fetch rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o r2 -- python $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_write.log 2>&1); echo "pmc write rc=$?"
