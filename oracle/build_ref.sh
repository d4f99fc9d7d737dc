#!/bin/sh
# Builds, from the reference's own sources where they lie under /root/reference (nothing is copied), plus the C-ABI shims
# in this directory:
#   oracle/_ref/libmod2mat_ref.so  <- lib/data/MNC/radford/mod2mat.cpp           (GF(2) matrices: the encoder of `s2t`)
#   oracle/_ref/libbnd_ref.so      <- lib/data/MNC/{zb2x.cpp, bnd/bnd.cpp, ansi/{cmatrix,nrutil,r,rand2}.cpp}
#                                                                                 (MacKay's sum-product decoder, `zb2x`)
# Test infrastructure only.  The pybind11/xtensor binding file MNC_py.cpp itself is NOT buildable here (no xtensor).
set -e
REF=${FGNN_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
MNC=$REF/lib/data/MNC
[ -f "$MNC/radford/mod2mat.cpp" ] || { echo "reference not present: $MNC" >&2; exit 3; }
mkdir -p "$HERE/_ref"
g++ -O2 -shared -fPIC -w -I"$MNC/radford" "$HERE/ref_mod2mat_shim.cpp" "$MNC/radford/mod2mat.cpp" -o "$HERE/_ref/libmod2mat_ref.so"
g++ -O2 -shared -fPIC -w -I"$MNC" -I"$MNC/ansi" "$HERE/ref_bnd_shim.cpp" "$MNC/zb2x.cpp" "$MNC/bnd/bnd.cpp" \
    "$MNC/ansi/cmatrix.cpp" "$MNC/ansi/nrutil.cpp" "$MNC/ansi/r.cpp" "$MNC/ansi/rand2.cpp" -o "$HERE/_ref/libbnd_ref.so"
echo "built $HERE/_ref/libmod2mat_ref.so $HERE/_ref/libbnd_ref.so"
