#!/bin/bash
# Build experiment variants of the library HERE (CPU container; hipcc cross-compiles) so that a GPU call only measures:
#   tools/build_variants.sh variants.txt          # each line: <tag> <-D flags...>
# -> 3dgsconverter_amd/variants/libgsx_hip_<tag>.so (travels to the GPU box; git-ignored)
set -u
cd "$(dirname "$0")/.."
while read -r tag flags; do
  [ -z "${tag:-}" ] && continue
  case "$tag" in \#*) continue;; esac
  ( GSX_VARIANT_TAG="$tag" GSX_EXTRA_FLAGS="$flags" GSX_VARIANT_ONLY="${ONLY:-sor_grid.hip}" python 3dgsconverter_amd/build.py > /tmp/build_$tag.log 2>&1 \
      && echo "built $tag: $flags" || { echo "FAILED $tag"; tail -5 /tmp/build_$tag.log; } ) &
  while [ "$(jobs -r | wc -l)" -ge "${JOBS:-4}" ]; do sleep 1; done
done < "$1"
wait
ls -la 3dgsconverter_amd/variants/
