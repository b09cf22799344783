#!/bin/bash
# Two PMC passes (FETCH_SIZE, WRITE_SIZE) of a short bench.py run + the recorder's family notes -> profiles/traffic[_cN].json
# usage (on the GPU box, from the repo root): bash scratch/pmc_traffic.sh c2|c4|c5
set -e
CFG=${1:-c2}
OUT=$PWD/gpurun_out/pmc_$CFG
mkdir -p $OUT
ARGS="--config $CFG --steps 3 --warmup 1 --no_cpu_baseline --full_steps 0 --eval_steps 1 --record_steps 0"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    B4C_FAMILY_LOG=$OUT/family_log_$C.json timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $ROOT/bench.py $ARGS > $OUT/bench_$C.json 2> $OUT/bench_$C.err
    echo "$CFG $C pass done"
done
cd $ROOT
case $CFG in
  c2) CONF='{"vocab": 50000, "batch": 4096, "seq": 200, "d_model": 128, "layers": 4, "dtype": "bf16"}'; NAME=traffic.json;;
  c4) CONF='{"vocab": 100000, "batch": 4096, "seq": 200, "d_model": 256, "layers": 6, "dtype": "bf16"}'; NAME=traffic_c4.json;;
  c5) CONF='{"vocab": 2000000, "batch": 1024, "seq": 512, "d_model": 256, "layers": 4, "dtype": "bf16"}'; NAME=traffic_c5.json;;
esac
python3 scratch/pmc_traffic.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/family_log_FETCH_SIZE.json $OUT/$NAME "$CONF" | tee $OUT/summary.txt
