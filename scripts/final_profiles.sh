#!/bin/bash
# Round-end measurement batch (run through gpurun): kernel-trace statistics of the cfg-3 and cfg-5 benches, the default bench line, the
# 30-sweep trajectory.  Outputs under gpurun_out/r4/.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4
cd $R
scripts/prof_ab.sh c3fin $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg5 --batched-chains 0 || exit 1
scripts/prof_ab.sh c5fin $R/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-cfg5 --batched-chains 0 || exit 1
cd $R && python bench.py > gpurun_out/r4/bench_final2.json 2> gpurun_out/r4/bench_final2.err || exit 1
python scripts/long_parity.py > gpurun_out/r4/long_parity2.log 2>&1
tail -2 gpurun_out/r4/long_parity2.log
