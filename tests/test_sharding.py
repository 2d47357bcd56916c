"""Multi-GPU sharding logic on CPU: partition arithmetic, and a world_size-2
gloo run in which each rank processes its shard (the oracle stands in for the
device kernels -- on the GPU box the default callables are the HIP engine) and
the all-gathered stream must equal the single-call result."""
import os
import socket

import pytest

import micro_aes_amd.sharding as sh


def test_shard_bounds():
    for total in (0, 1, 15, 16, 17, 4096, 1 << 20, (1 << 20) + 5):
        for world in (1, 2, 3, 8):
            b = sh.shard_bounds(total, world, 16)
            assert len(b) == world and sum(n for _, n in b) == total
            pos = 0
            for i, (s, n) in enumerate(b):
                assert s == pos and (n == 0 or s % 16 == 0)
                if s + n < total:
                    assert n % 16 == 0
                pos += n
    # XTS: whole data units only
    b = sh.shard_bounds(10 * 4096, 4, 4096)
    assert [n // 4096 for _, n in b] == [3, 3, 3, 1]
    assert sh.xts_shard_args(10, 4096, 100, 3, 4) == (9 * 4096, 1, 109)
    assert sh.ctr_shard_args(1 << 30, 3, 8) == (3 << 27, 1 << 27, 3 << 23)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    key, ctr0 = bytes(range(16)), bytes(range(0xF0, 0xFC)) + b"\x00\xff\xff\xfe"
    stream = orc.splitmix(2, total)
    cap = sh.shard_bounds(total, world, 16)[0][1]
    start, n, off = sh.ctr_shard_args(total, rank, world)
    src = torch.zeros(cap, dtype=torch.uint8)
    src[:n] = torch.frombuffer(bytearray(stream[start:start + n]), dtype=torch.uint8)
    dst = torch.zeros(cap, dtype=torch.uint8)

    def cpu_cipher(k, c, boff, s, d, nbytes):       # test stand-in for uaes_ctr_xcrypt_at_dev
        out = orc.ctr_xcrypt_at(k, c, boff, bytes(s[:nbytes].numpy()))
        d[:nbytes] = torch.frombuffer(bytearray(out), dtype=torch.uint8)

    gathered = torch.zeros(cap * world, dtype=torch.uint8)
    sh.ctr_xcrypt_sharded(key, ctr0, total, src, dst, rank, world, cipher=cpu_cipher, gather_into=gathered)
    got = b"".join(bytes(gathered[r * cap: r * cap + nn].numpy())
                   for r, (_, nn) in enumerate(sh.shard_bounds(total, world, 16)))
    want = orc.ctr_xcrypt_at(key, ctr0, 0, stream)

    # XTS volume of 7 data units of 528 bytes, sharded by whole units
    keys = bytes(range(64))
    vol = orc.splitmix(3, 7 * 528)
    xcap = sh.shard_bounds(7 * 528, world, 528)[0][1]
    xs, ns, first = sh.xts_shard_args(7, 528, 1000, rank, world)
    xsrc = torch.zeros(xcap, dtype=torch.uint8)
    xsrc[: ns * 528] = torch.frombuffer(bytearray(vol[xs: xs + ns * 528]), dtype=torch.uint8)
    xdst = torch.zeros(xcap, dtype=torch.uint8)

    def cpu_xts(k, f, sb, nsec, s, d, enc):
        rc, out = orc.xts_sectors(k, f, sb, bytes(s[: nsec * sb].numpy()), enc)
        assert rc == 0
        d[: nsec * sb] = torch.frombuffer(bytearray(out), dtype=torch.uint8)

    xg = torch.zeros(xcap * world, dtype=torch.uint8)
    sh.xts_sectors_sharded(keys, 1000, 528, 7, xsrc, xdst, rank, world, cipher=cpu_xts, gather_into=xg)
    xgot = b"".join(bytes(xg[r * xcap: r * xcap + nn].numpy())
                    for r, (_, nn) in enumerate(sh.shard_bounds(7 * 528, world, 528)))
    xwant = orc.xts_sectors(keys, 1000, 528, vol, True)[1]
    if rank == 0:
        q.put((got == want, xgot == xwant))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ctr_and_xts_sharded_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    total = 100000 + 7                 # ragged tail lands on the last rank
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    ok, t0 = None, time.time()
    while ok is None and time.time() - t0 < 240:
        try:
            ok = q.get(timeout=2)
        except queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died"
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok == (True, True)


def _oracle_partial(orc, key, nonce, aad, total_aad_len, ct, n, start, total):
    """CPU stand-in for uaes_gcm_partial_dev, from the oracle's primitives"""
    H = orc.encrypt_block(key, bytes(16))
    first, last = start == 0, start + n == total
    a_blk, c_blk = (total_aad_len + 15) // 16, (total + 15) // 16
    m_total = a_blk + c_blk + 1
    blocks = []
    if first and aad:
        blocks += [aad[i:i + 16].ljust(16, b"\0") for i in range(0, len(aad), 16)]
    blocks += [ct[i:i + 16].ljust(16, b"\0") for i in range(0, n, 16)]
    if last:
        blocks.append((total_aad_len * 8).to_bytes(8, "big") + (total * 8).to_bytes(8, "big"))
    hi = a_blk + (start + n + 15) // 16 + (1 if last else 0)
    acc = bytes(16)
    for b in blocks:
        acc = orc.gf128_mul(bytes(x ^ y for x, y in zip(acc, b)), H)
    e, p = m_total - hi, H
    one = bytes([0x80] + [0] * 15)
    w = one
    while e:
        if e & 1:
            w = orc.gf128_mul(w, p)
        p = orc.gf128_mul(p, p)
        e >>= 1
    acc = orc.gf128_mul(acc, w)
    if first:
        j0 = nonce + b"\0\0\0\1"
        acc = bytes(x ^ y for x, y in zip(acc, orc.encrypt_block(key, j0)))
    return acc


@pytest.mark.parametrize("total,world", [(0, 2), (5, 3), (16, 2), (1000, 3), (4096 + 7, 4), (100000, 8)])
def test_sharded_gcm_math(orc, total, world):
    """tag = XOR of per-shard weighted partial GHASHes == single-call GCM (oracle stand-ins, no GPU)"""
    import random
    rnd = random.Random(total + world)
    key, nonce, aad = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(rnd.choice([0, 3, 16, 40]))
    data = orc.splitmix(total + 1, total)
    want = orc.gcm_encrypt(key, nonce, aad, data)
    pieces, shares = [], []

    def cipher(k, c, off, s, d, n):
        d[:n] = orc.ctr_xcrypt_at(k, c, off, bytes(s[:n]))

    def partial(k, no, a, ta, ct, n, start, tot):
        return _oracle_partial(orc, k, no, a, ta, bytes(ct[:n]), n, start, tot)

    for rank in range(world):
        start, n, _ = sh.gcm_shard_roles(total, rank, world)
        src, dst = bytearray(data[start:start + n]), bytearray(n)
        # gather stand-in: collect this rank's share, return what is known so far
        tag = sh.gcm_encrypt_sharded(key, nonce, aad, len(aad), total, src, dst, rank, world,
                                     cipher=cipher, partial=partial,
                                     gather=lambda share: shares.append(share) or list(shares))
        pieces.append(bytes(dst))
    assert b"".join(pieces) == want[:-16]
    assert tag == want[-16:]            # after the last rank every share has been collected


@pytest.mark.parametrize("total,world", [(0, 2), (16, 3), (1000, 3), (100000, 8)])
def test_sharded_gcm_decrypt_authenticates_before_any_rank_writes(orc, total, world):
    """gcm_decrypt_sharded with oracle stand-ins: phase 1 = every rank's share over the RECEIVED ciphertext, phase 2 =
    the CTR pass, only when the XOR of the shares equals the tag; a forgery returns 0x1A on every rank with every
    local output untouched (N7 across ranks, micro_aes.c:1200-1208)"""
    import random
    rnd = random.Random(7 * total + world)
    key, nonce, aad = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(rnd.choice([0, 5, 32]))
    data = orc.splitmix(total + 3, total)
    msg = orc.gcm_encrypt(key, nonce, aad, data)

    def cipher(k, c, off, s, d, n):
        d[:n] = orc.ctr_xcrypt_at(k, c, off, bytes(s[:n]))

    def partial(k, no, a, ta, ct, n, start, tot):
        return _oracle_partial(orc, k, no, a, ta, bytes(ct[:n]), n, start, tot)

    for forged in (False, True):
        ct, tag = bytearray(msg[:-16]), bytearray(msg[-16:])
        if forged:
            if total and rnd.random() < 0.7:
                ct[rnd.randrange(total)] ^= 0x20
            else:
                tag[3] ^= 1
        # every rank's share first (what the all-gather would deliver), then every rank's call sees all of them
        shares = []
        for rank in range(world):
            start, n, takes = sh.gcm_shard_roles(total, rank, world)
            shares.append(partial(key, nonce, aad if rank == 0 else None, len(aad), ct[start:start + n], n, start, total)
                          if takes else bytes(16))
        outs = []
        for rank in range(world):
            start, n, _ = sh.gcm_shard_roles(total, rank, world)
            src, dst = bytearray(ct[start:start + n]), bytearray(b"\xAB" * n)
            rc = sh.gcm_decrypt_sharded(key, nonce, aad, len(aad), total, bytes(tag), src, dst, rank, world,
                                        cipher=cipher, partial=partial, gather=lambda share: list(shares))
            assert rc == (0x1A if forged else 0)
            outs.append(bytes(dst))
        assert b"".join(outs) == (b"\xAB" * total if forged else data)


def test_sharded_gcm_callables_default_independently(orc, monkeypatch):
    """`cipher` and `partial` default one by one.  A caller that injects only ONE of them keeps it -- the
    other becomes the engine's (here: stand-ins patched over sharding._dev_cipher / _dev_partial, no GPU) -- in both
    directions; encryption used to crash on the missing one, decryption used to throw the supplied one away."""
    key, nonce, aad, total, world = bytes(range(16)), bytes(range(12)), b"hdr", 1000, 2
    data = orc.splitmix(3, total)
    msg = orc.gcm_encrypt(key, nonce, aad, data)
    calls = {"my_cipher": 0, "my_partial": 0, "dev_cipher": 0, "dev_partial": 0}

    def make(name, fn):
        def wrapped(*a):
            calls[name] += 1
            return fn(*a)
        return wrapped

    def cipher(k, c, off, s, d, n):
        d[:n] = orc.ctr_xcrypt_at(k, c, off, bytes(s[:n]))

    def partial(k, no, a, ta, ct, n, start, tot):
        return _oracle_partial(orc, k, no, a, ta, bytes(ct[:n]), n, start, tot)

    monkeypatch.setattr(sh, "_dev_cipher", make("dev_cipher", cipher))
    monkeypatch.setattr(sh, "_dev_partial", make("dev_partial", partial))
    for kw in ({"cipher": make("my_cipher", cipher)}, {"partial": make("my_partial", partial)}):
        for k in calls:
            calls[k] = 0
        shares, pieces = [], []
        for rank in range(world):
            start, n, _ = sh.gcm_shard_roles(total, rank, world)
            src, dst = bytearray(data[start:start + n]), bytearray(n)
            tag = sh.gcm_encrypt_sharded(key, nonce, aad, len(aad), total, src, dst, rank, world,
                                         gather=lambda share: shares.append(share) or list(shares), **kw)
            pieces.append(bytes(dst))
        assert b"".join(pieces) + tag == msg
        mine, other = ("my_cipher", "dev_partial") if "cipher" in kw else ("my_partial", "dev_cipher")
        assert calls[mine] == world and calls[other] == world and calls["dev_cipher" if "cipher" in kw else "dev_partial"] == 0, calls
        for k in calls:
            calls[k] = 0
        outs = []
        for rank in range(world):
            start, n, _ = sh.gcm_shard_roles(total, rank, world)
            src, dst = bytearray(msg[start:start + n]), bytearray(n)
            assert sh.gcm_decrypt_sharded(key, nonce, aad, len(aad), total, msg[-16:], src, dst, rank, world,
                                          gather=lambda share: list(shares), **kw) == 0
            outs.append(bytes(dst))
        assert b"".join(outs) == data
        assert calls[mine] == world and calls[other] == world, calls


@pytest.mark.parametrize("total,world", [(0, 2), (5, 3), (16, 2), (4096 + 7, 4), (100000, 8)])
def test_sharded_ecb(orc, total, world):
    """ecb_shard_args / ecb_sharded: whole blocks per rank, the rank that holds the end of the text takes the ragged
    tail (N1: zero padded into a whole block); the concatenation is the one-call result"""
    key = bytes(range(16))
    data = orc.splitmix(total + 11, total)
    want = orc.ecb_encrypt(key, data)
    out, lasts = b"", 0

    def cipher(k, s, d, n, dec):
        d[: (n + 15) // 16 * 16] = orc.ecb_encrypt(k, bytes(s[:n]))

    for rank in range(world):
        start, n, last = sh.ecb_shard_args(total, rank, world)
        lasts += last
        assert n % 16 == 0 or last
        src, dst = bytearray(data[start:start + n]), bytearray((n + 15) // 16 * 16)
        assert sh.ecb_sharded(key, total, src, dst, rank, world, cipher=cipher) == (start, n)
        out += bytes(dst)
    assert out == want and lasts == 1


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts its own N ranks the way the documented
    torch.distributed.run command does (rendezvous on 127.0.0.1, a free port, the same arguments)."""
    import subprocess
    import sys
    import types
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                     # the ranks' status is ours
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_reports_the_gather_by_default_with_more_than_one_rank():
    """the driver's command is `bench.py --gpus N --steps K --warmup W` and nothing else -- with N > 1
    that alone must produce everything SURVEY.md 8e lists (gather, encrypt + gather, the digest of the concatenated
    stream, the C host's own gather); the one-GPU line stays as it is; --no-gather / --no-c-gather opt out"""
    import bench
    ap = bench.build_parser()
    a = ap.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "3"])
    assert a.gather is None and a.c_gather is None
    assert bench.resolve_gather_flags(a, 8) == (True, True)
    assert bench.resolve_gather_flags(a, 2) == (True, True)
    assert bench.resolve_gather_flags(ap.parse_args([]), 1) == (False, False)
    assert bench.resolve_gather_flags(ap.parse_args(["--gpus", "8", "--no-gather"]), 8) == (False, False)
    assert bench.resolve_gather_flags(ap.parse_args(["--gpus", "8", "--no-c-gather"]), 8) == (True, False)
    assert bench.resolve_gather_flags(ap.parse_args(["--gather", "--c-gather"]), 1) == (True, True)


def test_bench_bounded_wait_returns_an_answer_whatever_happens():
    import time
    import bench
    assert bench.bounded(lambda: 41 + 1, 5) == (42, None, False)
    res, err, alive = bench.bounded(lambda: 1 / 0, 5)
    assert res is None and err.startswith("ZeroDivisionError") and not alive
    res, err, alive = bench.bounded(lambda: time.sleep(3), 0.2)
    assert res is None and err.startswith("no answer within") and alive


def _collectives_worker(rank, world, port, q):
    """bench.setup_collectives without a GPU: the RCCL bring-up cannot work here (no device), so every rank must
    agree on the gloo fallback -- with the reason in `collective_backend` -- and the group handed back must work"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      UAES_BENCH_RCCL_WAIT_S="20")
    if rank == 1:
        os.environ["UAES_BENCH_FORCE_NCCL_FAIL"] = "1"        # ONE rank failing must be enough for all to fall back
    import types
    import torch
    import bench
    a = types.SimpleNamespace(backend="nccl")
    dist, group, cdev, info = bench.setup_collectives(a, torch, rank, world, 0, torch.device("cpu"))
    t = torch.tensor([rank + 1.0], device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dist.barrier(group=group)
    q.put((rank, group is None, str(cdev), info["collective_backend"], float(t.item())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bench_collectives_fall_back_to_gloo_when_rccl_does_not_come_up():
    """the first real N > 1 run must not be lost to an RCCL problem.  Two gloo ranks on the CPU: rank 1's
    bring-up is forced to fail, rank 0's fails for real (no GPU here); both must report the gloo fallback and the
    control plane (barrier + reductions) must work on what setup_collectives returned."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_collectives_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, is_default_group, cdev, backend, total in got:
        assert is_default_group and cdev == "cpu" and total == 3.0
        assert backend.startswith("gloo (nccl init failed: "), backend
    assert "forced by UAES_BENCH_FORCE_NCCL_FAIL" in got[1][3]
