#!/bin/bash
# plain bench against the same under torch.distributed.run (one rank), same box; SKINRT_FIX_STREAM=0 = the skin stage's kernels one after the other
run_plain() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s4_tr_$1.json 2> gpurun_out/s4_tr_$1.err || exit 1; }
run_torch() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-vertex-gather > gpurun_out/s4_tr_$1.json 2> gpurun_out/s4_tr_$1.err || exit 1; }
run_plain plain
SKINRT_FIX_STREAM=0 run_plain plain_serial
run_torch torch 29521
SKINRT_FIX_STREAM=0 run_torch torch_serial 29522
python3 tools/show_bench.py gpurun_out/s4_tr_plain.json gpurun_out/s4_tr_plain_serial.json gpurun_out/s4_tr_torch.json gpurun_out/s4_tr_torch_serial.json | cut -c1-230
