#!/usr/bin/env bash
# Build libphl.so, stop on any compile error, then run the given command on the MI355X box.
# usage: tools/gpu.sh [--timeout S] -- '<command>'
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
if ! make -C "$ROOT/depth-estimation_amd/csrc" -j4 > /tmp/phl_make.log 2>&1; then grep -E "error" -A6 /tmp/phl_make.log >&2; echo "BUILD FAILED" >&2; exit 1; fi
grep -E "warning" -A3 /tmp/phl_make.log || true
test "$ROOT/depth-estimation_amd/lib/libphl.so" -nt "$ROOT/depth-estimation_amd/csrc/phl_tiles.hip" || { echo "libphl.so older than sources: BUILD FAILED" >&2; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
