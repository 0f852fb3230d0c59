# A/B runs of bench.py's timed configuration under launch-shape knobs (debug library): one line per run
run() { python bench.py --warmup 5 --no-cpu-baseline --no-early-stop --no-other-configs $1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=l['roofline']
print('%-64s value %.1f b8 %.1f decode %.3f enc %.3f | xattn alone %.1f us frac %.3f, in situ %.1f us' % ('$1', l['value'], l.get('value_batch8') or 0, l['stage_roofline']['decode']['frac'], l['stage_roofline']['encoder_xkv']['frac'], r['avg_us'], r['frac'], (r.get('in_situ') or {}).get('avg_us',0)))"; }
for a in "$@"; do run "$a"; done
