#!/bin/bash
# the determinism test fails only AFTER the CLI test in the same process: which switch matters?
run() { echo "== $1: $(env $1 python -m pytest tests -q -x -m gpu -k 'train_from_folder_on_gpu or bit_identical_with_streams_on' 2>&1 | tail -1)"; }
for v in "$@"; do run "$v"; done
