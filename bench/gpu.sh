#!/bin/bash
# Local helper (build container): stamps the tree's git sha into .build_sha (travels with the snapshot; .git does
# not) and forwards to gpurun.  Usage: bench/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." && (git rev-parse --short HEAD; git status --short | grep -v '^??' | wc -l) | tr '\n' ' ' > .build_sha
exec /usr/local/graft/bin/gpurun "$@"
