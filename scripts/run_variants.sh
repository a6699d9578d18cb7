#!/bin/bash
# Run ON THE GPU BOX: for every library variant scripts/_variants/lib_<tag>.so (built beforehand with MI_EXTRA_HIPCC_FLAGS)
# copy it over the package's library in this scratch copy of the repo and run the given command; the shipped library is
# restored at the end.  Usage: scripts/run_variants.sh <command...>   (output: one block per variant on stdout)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$REPO/armadillocudalinearinterpolation_amd/libmi355interp.so
cp "$LIB" "$LIB.shipped"
for v in "$REPO"/scripts/_variants/lib_*.so; do
    t=$(basename "$v" .so); t=${t#lib_}
    cp "$v" "$LIB"
    echo "== variant $t"
    "$@" 2>&1 | grep -v amdgpu.ids
done
cp "$LIB.shipped" "$LIB"; rm -f "$LIB.shipped"
