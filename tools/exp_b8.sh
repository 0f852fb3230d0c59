# the literal configs[3] job (8 chunks, 224 tokens) as one group of 8 vs two lanes of 4 vs three lanes (3/3/2)
run() { python bench.py --warmup 2 --no-cpu-baseline --no-early-stop --no-other-configs --no-single-batch $1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %.1f ms/step %.1f stage %s' % ('$1', l['value'], l['ms_per_step'], {k: round(v,1) for k,v in l['stage_ms'].items()}))"; }
for a in "$@"; do run "$a"; done
