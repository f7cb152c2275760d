#!/bin/bash
# Compares the gfx950 device ISA of the kernel sources between a git revision and the working tree (no GPU needed):
#   profiles/isa_diff.sh [REV=HEAD] [file.hip ...]
# A refactor that should not change code generation must print "0 differing lines" for every file; only the per-compile
# __hip_cuid_* symbol is ignored.  Used for the km_regtile.h extraction (identical ISA for all seven files).
set -e
REV=${1:-HEAD}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FILES=${@:-$(cd "$ROOT/kornia_amd/csrc" && ls *.hip)}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -S"
TMP=$(mktemp -d)
mkdir -p "$TMP/old" "$TMP/a" "$TMP/b"
(cd "$ROOT" && git archive "$REV" kornia_amd/csrc | tar -x -C "$TMP/old")
for f in $FILES; do
  [ -f "$TMP/old/kornia_amd/csrc/$f" ] || { echo "$f: new file"; continue; }
  /opt/rocm/bin/hipcc $FLAGS "$TMP/old/kornia_amd/csrc/$f" -o "$TMP/a/$f.s" 2>/dev/null &
  /opt/rocm/bin/hipcc $FLAGS "$ROOT/kornia_amd/csrc/$f" -o "$TMP/b/$f.s" 2>/dev/null &
  wait
  n=$(diff <(grep -v '^\s*;' "$TMP/a/$f.s" | grep -v '__hip_cuid_') <(grep -v '^\s*;' "$TMP/b/$f.s" | grep -v '__hip_cuid_') | grep -c '^[<>]' || true)
  echo "$f: $n differing lines"
done
rm -rf "$TMP"
