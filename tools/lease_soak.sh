#!/bin/bash
# a longer window: 3000 timed steps of the bench workload (about 25 s of GPU time), loss trajectory end points
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/soak
timeout -k 10 300 python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-kernel-rooflines --long-steps 0 2> gpurun_out/soak/bench.err | tail -1 > gpurun_out/soak/line.json
python -c "import json; d=json.load(open('gpurun_out/soak/line.json')); print('3000 steps:', d['value'], 'samples/s', d['ms_per_step'], 'ms/step, final loss', d['final_loss'])"
