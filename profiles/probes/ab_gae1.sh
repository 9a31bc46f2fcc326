run() { echo "== $1"; env $1 python bench.py --gae 1 --no-cpu-baseline --roofline-steps 0 2>&1 | tail -1 | python -c "import sys,json;l=sys.stdin.read();j=json.loads(l[l.index('{'):]);print(j['value'],j['ms_per_step'])"; }
run X=1
run STYLEX_UPLOAD_STREAM=0
run STYLEX_RES_GEMM=0
run STYLEX_MODCOEFF=0
run STYLEX_PAD_RGB=0
run STYLEX_CONV_PIPE=0
run "STYLEX_UPLOAD_STREAM=0 STYLEX_RES_GEMM=0 STYLEX_MODCOEFF=0 STYLEX_PAD_RGB=0 STYLEX_CONV_PIPE=0"
run X=1
