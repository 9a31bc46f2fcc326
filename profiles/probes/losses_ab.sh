python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "fused_loss or trainer_step or trajectory or gp_and_pl or config4 or determinis" 2>&1 | tail -8
BENCH_ARGS="--steps 30 --warmup 8" bash tools/probes/ab_env.sh STYLEX_FUSED_LOSSES=0 STYLEX_FUSED_LOSSES=0
