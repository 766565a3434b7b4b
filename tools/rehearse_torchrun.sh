#!/bin/bash
# bench.py as the driver starts it for N > 1 (torch.distributed.run, RCCL), rehearsed with ONE rank on a one-GPU box:
# C3, and C4's N = 8 shard size (12,500 instances) with the bounds and vertex all-gathers.  Lines -> gpurun_out/r2_torchrun1*.json
set -u
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_torchrun1.json 2> gpurun_out/r2_torchrun1.err || { echo "c3 failed"; tail -5 gpurun_out/r2_torchrun1.err; exit 1; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload c4 --instances 12500 --steps 10 --warmup 3 > gpurun_out/r2_torchrun1_c4.json 2> gpurun_out/r2_torchrun1_c4.err || { echo "c4 failed"; tail -5 gpurun_out/r2_torchrun1_c4.err; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/r2_torchrun1.json", "gpurun_out/r2_torchrun1_c4.json"):
    d = json.loads(open(f).read())
    print(d["config"]["workload"][:2], "n_gpus", d["n_gpus"], "rccl_ranks", d["rccl_ranks"], "ms/step", round(d["ms_per_step"], 3), d.get("exchange"))
PY
