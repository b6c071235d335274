#!/bin/bash
# usage: bash scripts/r3_run.sh "<pytest args>" <script> [args...]: tests first (stop on failure), then one bench script
timeout -k 10 250 python -m pytest $1 -m gpu -q -x > gpurun_out/run_tests.log 2>&1; rc=$?; tail -3 gpurun_out/run_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; grep -n "^E " gpurun_out/run_tests.log | head -5; exit $rc; fi
shift
timeout -k 10 400 python "$@" 2>&1 | grep -v amdgpu.ids
