"""One line of bench.py's JSON in short: headline value, step time, roofline fraction, greedy extras."""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f %s, %.2f ms per step, roofline frac %.3f' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac']))
for k in ('greedy_b1', 'greedy_b3'):
    g = d.get('extra', {}).get(k)
    if g:
        print('%s: %.1f us per step, %.0f frames/s, frac %.3f (%s join tiles); batch %.0f frames/s' % (
            k, g['us_per_step'], g['frames_per_s'], g['roofline']['frac'], g.get('join_tiles'), g['batch']['frames_per_s']))
