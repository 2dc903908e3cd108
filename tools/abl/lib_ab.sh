#!/bin/bash
# same-box A/B of two builds of the library: $1 = the alternative .so (in-tree path), rest = the command to run under each
cd $GRAFT_REPO_ROOT
alt=$1; shift
cp item_alignment_amd/libitemalign_hip.so /tmp/lib_default.so
for rep in 1 2; do
  echo "== default build (rep $rep)"; "$@"
  cp $alt item_alignment_amd/libitemalign_hip.so
  echo "== $alt (rep $rep)"; "$@"
  cp /tmp/lib_default.so item_alignment_amd/libitemalign_hip.so
done
