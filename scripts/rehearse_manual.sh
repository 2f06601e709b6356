#!/bin/bash
# two bench.py ranks on cuda:0 over gloo, started by hand (no torchrun) so that each rank's stderr is kept and a failing
# rank never gets its peer killed in the middle of a kernel: the peer waits in a gloo collective and ends at its time limit
set -o pipefail
mkdir -p gpurun_out
export JF_BENCH_REHEARSE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 WORLD_SIZE=2 JF_BENCH_SECONDARY_TIMEOUT_S=60
ARGS="--gpus 2 --log-n ${1:-18} --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-prove --ultra-log-n 0 ${2:-}"
RANK=1 LOCAL_RANK=1 timeout -k 5 90 python -u bench.py $ARGS > gpurun_out/rm_rank1.out 2> gpurun_out/rm_rank1.err &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 5 90 python -u bench.py $ARGS > gpurun_out/rm_rank0.out 2> gpurun_out/rm_rank0.err
RC0=$?
wait $P1
RC1=$?
echo "rank0 rc=$RC0 rank1 rc=$RC1"
tail -c 1500 gpurun_out/rm_rank0.err; echo; tail -c 800 gpurun_out/rm_rank1.err; echo
tail -c 1200 gpurun_out/rm_rank0.out
[ $RC0 -eq 0 ] && [ $RC1 -eq 0 ]
