#!/bin/bash
# Stage the reference checkout for ONE end-to-end GPU session (tools/seams_e2e_gpu.sh): the GPU box has conda + astropy
# but no lightkurve, and the only channel to it is the repo snapshot.  The tarball goes under .stage/, which is
# git-ignored (it travels with `gpurun`, it is never committed — the reference sources are not vendored) and is deleted
# again by this script's `clean` mode.
#     bash tools/stage_reference.sh          # here, in the container that has /root/reference
#     gpurun -- 'bash tools/seams_e2e_gpu.sh'
#     bash tools/stage_reference.sh clean
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "clean" ]; then rm -rf .stage/lkref.tar.gz; echo cleaned; exit 0; fi
REF=${LK_REFERENCE_ROOT:-/root/reference}
mkdir -p .stage
( cd "$REF" && tar -czf "$OLDPWD/.stage/lkref.tar.gz" src/lightkurve tests/*.py tests/correctors tests/data )
ls -la .stage/lkref.tar.gz
