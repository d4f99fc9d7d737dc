#!/bin/sh
# Builds, from the reference's own sources where they lie under /root/reference (nothing is copied), plus the C-ABI shims
# in this directory:
#   oracle/_ref/libmod2mat_ref.so  <- lib/data/MNC/radford/mod2mat.cpp           (GF(2) matrices: the encoder of `s2t`)
#   oracle/_ref/libbnd_ref.so      <- lib/data/MNC/{zb2x.cpp, bnd/bnd.cpp, ansi/{cmatrix,nrutil,r,rand2}.cpp}
#                                                                                 (MacKay's sum-product decoder, `zb2x`)
#   oracle/_ref/MNC<ext-suffix>.so <- lib/data/MNC/MNC_py.cpp + the files above, against the xtensor / xtl / xtensor-python headers
#                                     the reference vendors under lib/data/MNC/3rdparty and this image's pybind11 + numpy headers:
#                                     the reference's own Python module (`s2t`, `t2y`, `y2b`, `zb2x`, `init_seed`).  One g++ line,
#                                     not the reference's cmake.  Only oracle/make_t2y_golden.py imports it (t2y's RNG stream).
# Test infrastructure only.
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
PY=${PYTHON:-python3}
if PYB=$($PY -m pybind11 --includes 2>/dev/null) && NPI=$($PY -c 'import numpy; print(numpy.get_include())' 2>/dev/null); then
    SUF=$($PY -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')
    g++ -O1 -shared -fPIC -std=c++14 -w $PYB -I"$NPI" -I"$MNC" -I"$MNC/3rdparty/xtensor/include" \
        -I"$MNC/3rdparty/xtensor-python/include" -I"$MNC/3rdparty/xtl/include" \
        "$MNC/MNC_py.cpp" "$MNC/zb2x.cpp" "$MNC/bnd/bnd.cpp" "$MNC/ansi/cmatrix.cpp" "$MNC/ansi/nrutil.cpp" "$MNC/ansi/r.cpp" \
        "$MNC/ansi/rand2.cpp" "$MNC/radford/mod2mat.cpp" -o "$HERE/_ref/MNC$SUF"
    echo "built $HERE/_ref/MNC$SUF"
else
    echo "pybind11 / numpy headers not found: the reference's MNC Python module was not built (only make_t2y_golden.py needs it)" >&2
fi
