#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
cp mneslam_amd/libmneslam_hip.so /tmp/lib_orig.so
cp profiles/_variants/lib_prof.so mneslam_amd/libmneslam_hip.so
python profiles/render_phase_times.py > gpurun_out/render_phases.txt 2>&1
cp /tmp/lib_orig.so mneslam_amd/libmneslam_hip.so
cat gpurun_out/render_phases.txt
