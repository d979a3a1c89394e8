"""Phase times and every launch above a share of the step, from a bench.py JSON line: python tools/launches.py FILE [SHARE]"""
import json, sys
d = json.load(open(sys.argv[1]))
share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
print(d['config']['workload'], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['whole_multiply']['phase_ms'].items()})
for l in d['launches']:
    if l['ms'] > share * d['ms_per_step']:
        print(l)
