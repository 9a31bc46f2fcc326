# extra bench lines of the final state (GPU box): sanity run, config 4, GAE = 1, host share
cd $GRAFT_REPO_ROOT
T=${1:-r04_k}
python tools/bf16_sanity.py 300 > gpurun_out/${T}_bf16_sanity_300steps.txt 2>&1; tail -3 gpurun_out/${T}_bf16_sanity_300steps.txt
python bench.py --image-size 128 --classifier mobilenet --pl-every 16 --start-step 5024 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config4_128px_mobilenet_pl16.json
python bench.py --gae 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_bf16_benchA_gae1.json
python bench.py --host-share 8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_host_share8.json
for f in gpurun_out/${T}_bench_*.json; do python -c "
import json,sys;l=open(sys.argv[1]).read();j=json.loads(l[l.index('{'):]);print(sys.argv[1], j['value'], j['ms_per_step'], (j.get('roofline') or {}).get('kernel'))" $f; done
