for q in 12 20 28 36; do for k in 32 48 56; do
MIPT_LEAF_QUORUM=$q MIPT_KEEP_BUSY=$k timeout -k 10 200 python bench.py --scene random100k --spp 128 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || exit 1
python -c "
import json; d=json.load(open('/tmp/b.json')); print('quorum', $q, 'keep', $k, round(d['ms_per_step'],1), d['stage_ms_per_step']['generate'], d['stage_ms_per_step']['extend'])"
done; done
