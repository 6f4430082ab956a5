#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'   -- retries while the pod's GPU slots are busy (exit 3 / transient)
T=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$out" | tail -60
  exit $rc
done
echo "no GPU slot after 40 tries"; exit 3
