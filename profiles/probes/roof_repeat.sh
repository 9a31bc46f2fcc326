#!/bin/bash
# roofline.frac run-to-run on one box: bash tools/probes/roof_repeat.sh [n]
for i in $(seq 1 ${1:-3}); do python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json;l=sys.stdin.read();j=json.loads(l[l.index('{'):]);r=j['roofline'];print(j['value'],r['achieved'],r['frac'],r['launches'],r['avg_launch_ms'],r['per_layer']['mixed_frac'],{k:(v.get('achieved') if isinstance(v,dict) else v) for k,v in r.get('classes',{}).items()})"; done
