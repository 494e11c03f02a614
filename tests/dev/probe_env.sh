#!/bin/bash
# development helper (GPU box): run a probe script against variant libraries.  tests/dev/probe_env.sh <script.py> <tag[:ENV=val]>...
cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/corona-13_amd/csrc/variants
script=$1; shift
for t in "$@"; do
  lib=${t%%:*}; env=${t#*:}; [ "$env" = "$t" ] && env=""
  echo "== $lib $env"
  env $env CORONA_MI_LIB=$V/libcorona_mi_$lib.so timeout 300 python3 $script || echo "$t FAILED rc=$?"
done
