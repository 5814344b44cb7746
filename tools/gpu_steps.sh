#!/bin/bash
# Run the given steps ("name::seconds::command") one after another on the GPU box; a step that FAILS (test assertion,
# non-zero exit) does not stop the sequence, a step that TIMES OUT or is killed does (no further GPU work after a hang).
# Output of every step goes to gpurun_out/<name>.log; one status line per step on stdout.
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
for spec in "$@"; do
  name=${spec%%::*}; rest=${spec#*::}; secs=${rest%%::*}; cmd=${rest#*::}
  start=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $root/gpurun_out/$name.log 2>&1
  rc=$?
  echo "[$name] rc=$rc $(( $(date +%s) - start ))s"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping"; exit 1; fi
done
exit 0
