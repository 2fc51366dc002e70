#!/bin/bash
# diagnostic: repeat the bf16x3 tolerance tests (race screen; placement of reads is by vmcnt/barrier count, this only screens)
for i in 1 2 3 4; do
  timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bf16x3" 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' '; echo
done
