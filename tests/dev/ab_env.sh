cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/corona-13_amd/csrc/variants
for t in "$@"; do
  lib=${t%%:*}; env=${t#*:}; [ "$env" = "$t" ] && env=""
  echo "== $lib $env"
  env $env CORONA_MI_LIB=$V/libcorona_mi_$lib.so timeout 300 python3 tests/dev/ab_one.py || echo "$t FAILED rc=$?"
done
