#!/usr/bin/env python3
"""hipGraph capture of the *_dev entry points: they only enqueue kernels on the caller's stream (no
allocation after the first call, no synchronisation), so a launch-bound sequence of small calls can be
captured once and replayed.  Prints the per-call time of 64 x 64 KiB calls, direct and replayed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

key, nonce, keys2 = bytes(range(16)), bytes(12), bytes(range(64))
ctr0 = nonce + b"\0\0\0\1"
N, M = 64, 64 << 10
src = torch.randint(0, 256, (N, M), dtype=torch.uint8, device="cuda")
status = torch.zeros(1, dtype=torch.int32, device="cuda")


def sequence(name, dst, st):
    for i in range(N):
        if name == "ctr":
            uaes.ctr_xcrypt_dev(key, ctr0, i * (M // 16), src[i], dst[i, :M], nbytes=M, stream=st)
        elif name == "xts":
            uaes.xts_sectors_dev(keys2, i * (M // 4096), 4096, M // 4096, src[i], dst[i, :M], stream=st)
        elif name == "gcm":
            uaes.gcm_encrypt_dev(key, nonce, None, src[i], M, dst[i], stream=st)
        elif name == "ocb":
            uaes.ocb_dev(key, nonce, None, src[i], M, dst[i], stream=st)


for name in ("ctr", "xts", "gcm", "ocb"):
    direct = torch.zeros(N, M + 16, dtype=torch.uint8, device="cuda")
    replay = torch.zeros(N, M + 16, dtype=torch.uint8, device="cuda")
    sequence(name, direct, torch.cuda.current_stream())           # warm-up: contexts, scratch, attributes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        sequence(name, direct, torch.cuda.current_stream())
    torch.cuda.synchronize()
    t_direct = (time.perf_counter() - t0) / 10 / N
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sequence(name, replay, side)                              # this stream's scratch slot
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        sequence(name, replay, torch.cuda.current_stream())
    replay.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(direct, replay)
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 10 / N
    print("%-4s 64 x 64 KiB: direct %6.1f us/call, graph replay %6.1f us/call, identical output: %s"
          % (name, t_direct * 1e6, t_graph * 1e6, same))
