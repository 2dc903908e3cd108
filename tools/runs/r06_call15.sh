#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/runs/run.sh suite
