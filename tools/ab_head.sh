#!/bin/bash
# Same-box A/B of the working tree against a committed revision of the kernels (default HEAD).
#   here      : bash tools/ab_head.sh snapshot [rev]        -> build_ab/old_csrc/ (git-ignored, travels with gpurun)
#   on the box: gpurun -- 'bash tools/ab_head.sh run python tools/time_qp_dev.py 4,8,16,4096'
# `run` builds the snapshot in /tmp/anet_old (python layer and tools of the working tree, csrc of the revision), then runs the
# command OLD / NEW / OLD / NEW.  (If the working tree's ctypes table names entry points the old library lacks, the OLD runs fail to load.)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$1" in
snapshot)
  REV=${2:-HEAD}
  rm -rf "$ROOT/build_ab/old_csrc" && mkdir -p "$ROOT/build_ab/old_csrc"
  for f in $(git -C "$ROOT" ls-tree --name-only "$REV" allocnet_amd/csrc/); do git -C "$ROOT" show "$REV:$f" > "$ROOT/build_ab/old_csrc/$(basename $f)"; done
  echo "snapshot of $REV: $(ls $ROOT/build_ab/old_csrc | wc -l) files" ;;
run)
  shift
  rm -rf /tmp/anet_old && cp -r $ROOT /tmp/anet_old && cp $ROOT/build_ab/old_csrc/* /tmp/anet_old/allocnet_amd/csrc/
  (cd /tmp/anet_old && python -m allocnet_amd.build --force > /tmp/old_build.log 2>&1 || tail -5 /tmp/old_build.log)
  for k in 1 2; do
    echo "=== OLD"; (cd /tmp/anet_old && GRAFT_REPO_ROOT=/tmp/anet_old "$@" 2>&1 | grep -v amdgpu.ids)
    echo "=== NEW"; (cd $ROOT && "$@" 2>&1 | grep -v amdgpu.ids)
  done ;;
*) echo "usage: ab_head.sh snapshot [rev] | run <command>"; exit 2 ;;
esac
