#!/bin/bash
# per-call GPU job of the current experiment (edited per call; see git log for earlier forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
bash tools/profile_run.sh $TAG 2>&1 | tail -14
