#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "three_fused_mapping" 2>&1 | tail -60
MNE_WGRAD_INLINE=0 timeout 600 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "three_fused_mapping" 2>&1 | tail -5
