timeout 600 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "line64" 2>&1 | grep -v Warn | tail -15
echo NEW; python tools/bench_masks.py 2>/dev/null | grep -E "64->64|8->64" ; python tools/bench_conv.py --batch 64 --only D0.conv2,D1.conv1 2>/dev/null | grep -E "D0|D1"
echo OLD; STYLEX_CONV_LINE64=0 python tools/bench_masks.py 2>/dev/null | grep -E "64->64"; STYLEX_CONV_LINE64=0 python tools/bench_conv.py --batch 64 --only D0.conv2 2>/dev/null | grep -E "D0"
