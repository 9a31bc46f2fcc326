cd $GRAFT_REPO_ROOT
bash tools/trace_bench.sh gae1 --gae 1 --steps 24 --warmup 8 --roofline-steps 0 --fp32-steps 0
MS=$(python -c "import json;l=open('gpurun_out/trace_gae1_bench.txt').read();print(16*json.loads(l[l.index('{'):])['ms_per_step'])")
python tools/prof_window.py gpurun_out/trace_gae1.csv.gz $MS 16 40 2>&1 | head -45 | cut -c1-150
cat gpurun_out/trace_gae1_bench.txt | cut -c1-200
