#!/bin/bash
# scratch/gpu_retry.sh TIMEOUT 'command' : gpurun with retries while no slot / box is free (exit code 3)
T=$1; shift
for k in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
