#!/bin/bash
# One gpurun call: GPU tests (output kept under gpurun_out/), then the steps given as arguments; a timed-out or killed step ends the call.
# usage: bash scripts/gpu_call.sh <tag> [pytest-args ...] -- <cmd> [-- <cmd> ...]
mkdir -p gpurun_out
TAG=$1; shift
PYARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PYARGS+=("$1"); shift; done
if [ ${#PYARGS[@]} -gt 0 ]; then
  timeout -k 10 1000 python -m pytest "${PYARGS[@]}" > gpurun_out/${TAG}_tests.log 2>&1
  rc=$?; echo "pytest rc=$rc"; tail -n 6 gpurun_out/${TAG}_tests.log
  if [ $rc -ge 124 ]; then echo "tests timed out / were killed: stopping"; exit $rc; fi
fi
while [ $# -gt 0 ]; do
  shift   # the "--"
  CMD=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do CMD+=("$1"); shift; done
  [ ${#CMD[@]} -eq 0 ] && continue
  echo "== ${CMD[*]}"
  "${CMD[@]}"; rc=$?
  echo "rc=$rc"
  if [ $rc -ge 124 ]; then echo "step timed out / was killed: stopping"; exit $rc; fi
done
