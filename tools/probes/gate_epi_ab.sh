# gated epilogue of conv_pipe: new lib vs the previous build (libstylex_hip_prev.so), parity first
P=$GRAFT_REPO_ROOT/explaining-in-style-reproducibility-study_amd
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "pipelined or mask or dblock or DBlock or trainer_step or config2" 2>&1 | tail -3
export BENCH_ARGS="--steps 30 --warmup 8"
bash tools/probes/ab_env.sh STYLEX_HIP_LIB=$P/libstylex_hip_prev.so STYLEX_GATE_MASK_MIN_PIXELS=0 STYLEX_HIP_LIB=$P/libstylex_hip_prev.so STYLEX_GATE_MASK_MIN_PIXELS=0
