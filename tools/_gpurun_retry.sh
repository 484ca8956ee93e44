#!/usr/bin/env bash
# usage: tools/_gpurun_retry.sh <timeout> <command...>   -- retries while no GPU slot is free (nothing is charged for those attempts)
T="$1"; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$T" -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out" | tail -60; exit $rc
done
echo "gave up waiting for a GPU slot"; exit 3
