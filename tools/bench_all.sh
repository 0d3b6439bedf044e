#!/bin/bash
# The four bench lines (demo 1080p / 1024 spp, 100k at 256 spp, 20k mirrors / glass at 4K / 64 spp, 1M at 4K / 32 spp) into gpurun_out/<tag>_*.json.
# usage: tools/bench_all.sh <tag> [extra bench args...]
tag=$1; shift
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err || exit 1; python3 -c "
import json,sys; d=json.load(open('gpurun_out/${tag}_$name.json')); print('$name', round(d['ms_per_step'],1), 'ms', {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, 'roofline frac', round(d['roofline']['frac'],3))"; }
run demo1080p --steps 5 --warmup 1 "$@" &&
run random100k_256spp --scene random100k --spp 256 --steps 3 --warmup 1 "$@" &&
run mixed20k_4k_64spp --scene mixed20k --width 3840 --height 2160 --spp 64 --steps 3 --warmup 1 "$@" &&
run random1m_4k_32spp --scene random1m --width 3840 --height 2160 --spp 32 --steps 3 --warmup 1 "$@"
