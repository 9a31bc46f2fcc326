#!/bin/bash
# usage: bash tools/ab_env.sh "VAR=0" "VAR2=0" ...   -> bench.py with the default env, then with each override, then default again
run() { echo "== $1"; env $1 python bench.py --no-cpu-baseline --roofline-steps 0 --bench-a-steps 0 $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json;l=sys.stdin.read();j=json.loads(l[l.index('{'):]);print(j['value'],j['ms_per_step'])"; }
run X=1
for v in "$@"; do run "$v"; done
run X=1
