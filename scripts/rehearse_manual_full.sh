#!/bin/bash
# like rehearse_manual.sh but WITH the prove / batch_prove sections (2^20-gate TurboPlonk sharded over the two ranks, one
# 2^16-gate UltraPlonk circuit per rank): the N > 1 code path of every bench section on the one-GPU box
set -o pipefail
mkdir -p gpurun_out
export JF_BENCH_REHEARSE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 JF_BENCH_SECONDARY_TIMEOUT_S=280
ARGS="--gpus 2 --log-n 18 --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --ultra-log-n 16"
RANK=1 LOCAL_RANK=1 timeout -k 5 300 python -u bench.py $ARGS > gpurun_out/rmf_rank1.out 2> gpurun_out/rmf_rank1.err &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 5 300 python -u bench.py $ARGS > gpurun_out/rmf_rank0.out 2> gpurun_out/rmf_rank0.err
RC0=$?
wait $P1
RC1=$?
echo "rank0 rc=$RC0 rank1 rc=$RC1"
tail -c 800 gpurun_out/rmf_rank0.err; echo; tail -c 500 gpurun_out/rmf_rank1.err; echo
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/rmf_rank0.out").read().strip().splitlines()[-1])
print({k: (v.get("ms", v.get("error")) if isinstance(v, dict) else v) for k, v in d.items() if k in ("n_gpus", "value", "plonk_prove", "plonk_prove_hot_path", "batch_prove", "batch_prove_ultra", "weak_scaling")})
PY
[ $RC0 -eq 0 ] && [ $RC1 -eq 0 ]
