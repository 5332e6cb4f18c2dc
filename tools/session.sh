#!/bin/bash
# Run a list of GPU steps on the box, each under its own timeout, logging to gpurun_out/<name>.log; a step that times out
# or is killed ends the session (no further GPU step after a hang), an ordinary failure does not.
# usage: bash tools/session.sh name1 'cmd1' name2 'cmd2' ...
R=${GRAFT_REPO_ROOT:-.}
cd $R
export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  name=$1; cmd=$2; shift 2
  echo "== $name: $cmd" | tee gpurun_out/$name.log
  timeout -k 10 ${STEP_TIMEOUT:-600} bash -c "$cmd" >> gpurun_out/$name.log 2>&1
  rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/$name.log
  tail -n ${TAILN:-6} gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping"; exit 1; fi
done
exit 0
