#!/bin/bash
# tools/first_call_latency.sh -- VERDICT r04 #5: first-call latency of a process that uses two key sizes, with the thin
# shims of this round and with libraries linked the round-4 way (all engine objects in each).  Run on the GPU box:
#     gpurun -- 'bash tools/first_call_latency.sh'      (needs the objects of a finished `make` in csrc/build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
B=micro-aes_amd/csrc/build
FAT=$(mktemp -d /tmp/uaes_fat_XXXX)
for bits in 128 256; do
    gcc -O2 -std=gnu99 -fPIC -I/opt/rocm/include -DAES___=$bits -c micro-aes_amd/csrc/uaes_compat.c -o $FAT/compat_$bits.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $FAT/libmicro_aes_hip_$bits.so $FAT/compat_$bits.o \
        $B/uaes_kernels.o $B/uaes_gcm.o $B/uaes_mac.o $B/uaes_chain.o $B/uaes_ocb.o $B/uaes_engine.o $B/uaes_host.o -lpthread -ldl
done
gcc -O2 -o $FAT/first_call tools/first_call.c -ldl
ls -la micro-aes_amd/lib/*.so $FAT/*.so | awk '{print "  " $5, $9}'
for rep in 1 2 3; do
    echo "thin shims (this round: one engine, libmicro_aes_hip_<bits>.so = the adapter only), run $rep"
    $FAT/first_call micro-aes_amd/lib
    echo "fat libraries (round 4's link: every engine object in each per-key-size library), run $rep"
    $FAT/first_call $FAT
done
rm -rf $FAT
