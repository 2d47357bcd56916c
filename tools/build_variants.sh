#!/bin/bash
# tools/build_variants.sh <name>=<XFLAGS> ... -- builds libuaes_hip.so with the given extra hipcc flags into
# micro-aes_amd/lib/libuaes_hip_<name>.so (own object directory per variant, all in parallel; the product build in
# micro-aes_amd/lib/libuaes_hip.so is not touched).  For A/B runs with tools/ab_libs.py:
#     tools/build_variants.sh base= o2=-O2
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
pids=()
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    ( d=$(mktemp -d /tmp/uaes_var_XXXX)
      make -s -C "$ROOT/micro-aes_amd/csrc" -j6 OBJ=$d/obj OUT=$d/out XFLAGS="$flags" $d/out/libuaes_hip.so > $d/log 2>&1 \
        && cp $d/out/libuaes_hip.so "$ROOT/micro-aes_amd/lib/libuaes_hip_$name.so" && echo "built $name ($flags)" \
        || { echo "FAILED $name"; tail -20 $d/log; }
      rm -rf $d ) &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
