# in-step time of the 64 -> 64 @256^2 launches with and without conv_line64 (steady-state windows of a traced bench)
cd $GRAFT_REPO_ROOT
bash tools/trace_bench.sh l64on --steps 12 --warmup 6 --roofline-steps 0 --fp32-steps 0
STYLEX_CONV_LINE64=0 bash tools/trace_bench.sh l64off --steps 12 --warmup 6 --roofline-steps 0 --fp32-steps 0
for t in l64on l64off; do
  MS=$(python -c "import json;l=open('gpurun_out/trace_${t}_bench.txt').read();print(8*json.loads(l[l.index('{'):])['ms_per_step'])")
  echo "== $t"; python tools/prof_window.py gpurun_out/trace_$t.csv.gz $MS 8 70 2>&1 | grep -E "window|line64|pipe_kernel<64|halo_dma_kernel<2, false>|wgrad_halo_dma"
done
