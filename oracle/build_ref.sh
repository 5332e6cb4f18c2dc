#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Builds oracle/_ref/libphl_ref.so from the reference's own
# lattice engine where it lies under /root/reference (container only -- the
# reference does not exist on the GPU box; the built .so travels instead).
#
# Recipe: permutohedral.h lines 1-2, 4-198, 322-580, i.e.
# everything except `#include <torch/torch.h>` (line 3) and the at::Tensor
# marshalling `filter()` (:199-321) that does not compile on torch 2.10
# (`tensorFromBlob` was removed).  The filtered view lives in a mktemp dir that
# is deleted on exit: no reference source text is ever written into the repo.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF_H="${PHL_REFERENCE_ROOT:-/root/reference}/crf/lattice/lite/permutohedral.h"
if [ ! -f "$REF_H" ]; then
    echo "build_ref.sh: $REF_H not present (GPU box?) -- keeping prebuilt oracle/_ref" >&2
    exit 0
fi
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
sed -n '1,2p;4,198p;322,580p' "$REF_H" > "$TMP/perm_nofilter.h"
mkdir -p "$HERE/_ref"
# -O2, no -march / -ffast-math: plain IEEE fp32, no FMA contraction on x86-64.
g++ -O2 -w -fPIC -shared -I"$TMP" "$HERE/ref_harness.cpp" -o "$HERE/_ref/libphl_ref.so"
echo "built $HERE/_ref/libphl_ref.so"
