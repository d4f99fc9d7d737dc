#!/bin/sh
# Builds oracle/_ref/libmod2mat_ref.so from the reference's own GF(2) matrix source (where it lies under
# /root/reference — nothing is copied) plus the C-ABI shim oracle/ref_mod2mat_shim.cpp.  Test infrastructure only.
set -e
REF=${FGNN_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$REF/lib/data/MNC/radford
[ -f "$SRC/mod2mat.cpp" ] || { echo "reference not present: $SRC" >&2; exit 3; }
mkdir -p "$HERE/_ref"
g++ -O2 -shared -fPIC -w -I"$SRC" "$HERE/ref_mod2mat_shim.cpp" "$SRC/mod2mat.cpp" -o "$HERE/_ref/libmod2mat_ref.so"
echo "built $HERE/_ref/libmod2mat_ref.so"
