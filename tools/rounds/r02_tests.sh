#!/bin/bash
# GPU parity tests (+ optional extra command) -> gpurun_out/$1
out=gpurun_out/${1:-r02_t}; mkdir -p $out; shift
timeout 1200 python -m pytest tests -m gpu -x -q -s "$@" > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
grep -E "max abs diff|passed|failed|Error|error|assert" $out/pytest.log | tail -40
