#!/bin/bash
# Scheduling parameters of the one-ray-per-lane traversal kernels on the 100k-object scene: frame, primary and k_extend milliseconds for
#   MIPT_LEAF_QUORUM_PRIMARY (camera rays), then MIPT_LEAF_QUORUM x MIPT_KEEP_BUSY (bounce rays; MIPT_PAIR_QUORUM=0 selects that kernel --
#   the default for bounce rays is two rays per lane, swept by tools/sweep_pair.py).
run() {
    timeout -k 10 200 python bench.py --scene random100k --spp 128 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || exit 1
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('$1', round(d['ms_per_step'],1), d['stage_ms_per_step']['generate'], d['stage_ms_per_step']['extend'])"
}
for q in 2 4 8 12 16; do MIPT_LEAF_QUORUM_PRIMARY=$q run "primary quorum $q"; done
for q in 12 20 28; do for k in 32 40 48; do
MIPT_PAIR_QUORUM=0 MIPT_LEAF_QUORUM=$q MIPT_KEEP_BUSY=$k run "quorum $q keep $k"
done; done
