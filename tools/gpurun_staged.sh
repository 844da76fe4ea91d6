#!/bin/bash
# One gpurun call with the unmodified reference package staged next to the repo (tools/stage_reference.py): the scratch copy
# `_stage/` exists only for the duration of the call -- nothing of the reference stays in the tree.
#   tools/gpurun_staged.sh [--timeout S] -- '<command run on the GPU box>'
HERE="$(cd "$(dirname "$0")" && pwd)"
python "$HERE/stage_reference.py" || exit 1
trap 'python "$HERE/stage_reference.py" --clean' EXIT
/usr/local/graft/bin/gpurun "$@"
