"""Sharding of the block-parallel modes across the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).
CTR, ECB and XTS are embarrassingly parallel over 16-byte blocks / data units,
so a stream is cut into contiguous, unit-aligned shards and every rank runs the
same single-GPU kernels on its own shard:

* CTR -- rank g starts its keystream at ``block_offset = shard_start / 16``;
  the offset is added to the initial counter with the reference's 56-bit
  big-endian carry (incBlock, micro_aes.c:421-427), so the concatenation of
  the shards is bit-identical to one ``AES_CTR_encrypt`` call over the whole
  stream (micro_aes.c:962).
* XTS -- shards are whole data units; rank g starts at ``first_sector +
  shard_start / sector_bytes`` (the sectid convention, micro_aes.c:1017-1021).
* ECB -- any block partition.

* GCM -- CTR shards as above (counter block ``nonce || 00000001``, block offset
  ``1 + shard_start/16``: the CCM_GCM pre-increment, micro_aes.c:939-941); for
  the tag every rank computes its 16-byte share of ``Enc(J0) ^ GHASH`` over its
  own ciphertext shard (the shard's GHASH weighted by ``H^(blocks after the
  shard)``; the first shard carries the AAD and Enc(J0), the last one the length
  block) -- in the same pass as the CTR (``uaes_gcm_shard_dev``) -- and the
  shares are XORed after a 16-byte-per-rank all-gather: the one real exchange
  step on this path.  Decryption keeps N7 (micro_aes.c:1200-1208) across ranks:
  shares over the received ciphertext first (``uaes_gcm_partial_dev``), the
  all-gather, the comparison on every rank, and only then the CTR pass.

(``uaes_mgpu_*`` in include/uaes_hip.h is the same partitioning done by the C
host inside ONE process; this module is the one-process-per-GPU form.)

There is NO collective on the bulk data path.  ``gather`` is the optional final
step north_star asks for (ciphertext all-gather over xGMI); it costs ~20x the
encrypt itself (7 peers x 1 GiB over ~153 GB/s links vs. ~0.5 ms of kernel), so
callers that keep data sharded should leave it off.

The cipher is injected as a callable so that the partition/offset logic can be
exercised on CPU (gloo, world_size 2) in tests with the oracle standing in for
the device kernels; the default callables are the HIP engine's ``*_dev`` entry
points.
"""


def shard_bounds(total_bytes, world, unit=16):
    """Contiguous shards, each a multiple of `unit` except possibly the last.
    Returns [(start, nbytes)] of length `world` (empty shards allowed)."""
    units = (total_bytes + unit - 1) // unit
    per = (units + world - 1) // world
    out = []
    for r in range(world):
        start = min(r * per * unit, total_bytes)
        end = min((r + 1) * per * unit, total_bytes)
        out.append((start, end - start))
    return out


def ctr_shard_args(total_bytes, rank, world):
    """(byte_start, nbytes, block_offset) of rank's CTR shard."""
    start, n = shard_bounds(total_bytes, world, 16)[rank]
    return start, n, start // 16


def xts_shard_args(nsectors, sector_bytes, first_sector, rank, world):
    """(byte_start, nsectors_local, first_sector_local) of rank's XTS shard."""
    start, n = shard_bounds(nsectors * sector_bytes, world, sector_bytes)[rank]
    return start, n // sector_bytes, first_sector + start // sector_bytes


def ecb_shard_args(total_bytes, rank, world):
    """(byte_start, nbytes, is_last) of rank's ECB shard: whole blocks; the rank that holds the end of the text also
    takes the ragged tail and the padding (N1, padBlock micro_aes.c:610-621)."""
    bounds = shard_bounds(total_bytes, world, 16)
    start, n = bounds[rank]
    last = max((r for r in range(world) if bounds[r][1] > 0), default=0)
    return start, n, rank == last


def ecb_sharded(key, total_bytes, local_src, local_dst, rank, world, decrypt=False, cipher=None,
                gather_into=None, group=None):
    """Encrypt / decrypt this rank's block-aligned shard of an ECB text (AES_ECB_encrypt / _decrypt,
    micro_aes.c:636-680: any block partition).  cipher(key, src, dst, nbytes, decrypt) defaults to the HIP engine,
    which zero-pads a ragged tail on the rank that holds it (local_dst then receives nbytes rounded up to 16)."""
    if cipher is None:
        from . import ecb_dev

        def cipher(k, s, d, n, dec):
            ecb_dev(k, s, d, decrypt=dec, nbytes=n)
    start, n, _ = ecb_shard_args(total_bytes, rank, world)
    if n:
        cipher(key, local_src, local_dst, n, decrypt)
    if gather_into is not None:
        import torch.distributed as dist
        dist.all_gather_into_tensor(gather_into, local_dst, group=group)
    return start, n


def ctr_xcrypt_sharded(key, ctr0, total_bytes, local_src, local_dst, rank, world,
                       cipher=None, gather_into=None, group=None):
    """Encrypt this rank's shard of a `total_bytes` CTR stream.

    local_src/local_dst hold exactly this rank's shard.  cipher(key, ctr0,
    block_offset, src, dst, nbytes) defaults to the HIP engine.  If
    `gather_into` (a tensor of world * shard_capacity bytes) is given, the
    ciphertext shards are all-gathered into it afterwards (RCCL).
    """
    if cipher is None:
        from . import ctr_xcrypt_dev

        def cipher(k, c, off, s, d, n):
            ctr_xcrypt_dev(k, c, off, s, d, nbytes=n)
    start, n, off = ctr_shard_args(total_bytes, rank, world)
    if n:
        cipher(key, ctr0, off, local_src, local_dst, n)
    if gather_into is not None:
        import torch.distributed as dist
        dist.all_gather_into_tensor(gather_into, local_dst, group=group)
    return start, n


def xts_sectors_sharded(keys, first_sector, sector_bytes, nsectors, local_src, local_dst,
                        rank, world, encrypt=True, cipher=None, gather_into=None, group=None):
    """Encrypt/decrypt this rank's whole-data-unit shard of an XTS volume."""
    if cipher is None:
        from . import xts_sectors_dev

        def cipher(k, first, sb, ns, s, d, enc):
            xts_sectors_dev(k, first, sb, ns, s, d, encrypt=enc)
    start, ns, first = xts_shard_args(nsectors, sector_bytes, first_sector, rank, world)
    if ns:
        cipher(keys, first, sector_bytes, ns, local_src, local_dst, encrypt)
    if gather_into is not None:
        import torch.distributed as dist
        dist.all_gather_into_tensor(gather_into, local_dst, group=group)
    return start, ns


def gcm_shard_roles(total_len, rank, world):
    """(byte_start, nbytes, participates) of rank's GCM shard.  Rank 0 always
    participates (it owns the AAD / Enc(J0) and, for a short message, the length
    block); a later rank with an empty shard contributes nothing."""
    start, n = shard_bounds(total_len, world, 16)[rank]
    return start, n, (rank == 0 or n > 0)


def _xor_shares(shares):
    tag = bytes(16)
    for sh in shares:
        tag = bytes(a ^ b for a, b in zip(tag, sh))
    return tag


def _dev_cipher(k, c, off, s, d, n):
    from . import ctr_xcrypt_dev
    ctr_xcrypt_dev(k, c, off, s, d, nbytes=n)


def _dev_partial(k, no, a, ta, ct, n, start, total):
    import torch
    from . import gcm_partial_dev
    out = torch.zeros(16, dtype=torch.uint8, device=ct.device)
    gcm_partial_dev(k, no, a, ta, ct, n, start, total, out)
    return out


def _exchange(share, world, device, gather):
    """all ranks' 16-byte shares (this rank's may be None = zeros)"""
    if gather is None:
        import torch
        import torch.distributed as dist
        mine = share if share is not None else torch.zeros(16, dtype=torch.uint8, device=device)
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        return [bytes(t.cpu().numpy()) for t in allp]
    if share is None:
        mine = bytes(16)
    elif hasattr(share, "cpu"):
        mine = bytes(share.cpu().numpy())
    else:
        mine = bytes(share)
    return gather(mine)


def gcm_encrypt_sharded(key, nonce, aad, total_aad_len, total_len, local_src, local_dst, rank, world,
                        cipher=None, partial=None, gather=None):
    """Encrypt this rank's shard of ONE GCM message and return the 16-byte tag
    (identical on every rank).

    cipher(key, ctr0, block_offset, src, dst, nbytes) and
    partial(key, nonce, aad, total_aad_len, ct_shard, nbytes, shard_start, total_len) -> 16 bytes
    are the injectable two-step form (CPU tests); with neither given the HIP engine does both in ONE pass
    (uaes_gcm_shard_dev).  gather(share: bytes) -> list of all ranks' shares
    defaults to torch.distributed.all_gather (RCCL over xGMI).
    """
    start, n, takes_part = gcm_shard_roles(total_len, rank, world)
    if cipher is None and partial is None:
        import torch
        from . import gcm_shard_dev
        share = None
        if takes_part:
            share = torch.zeros(16, dtype=torch.uint8, device=local_dst.device)
            gcm_shard_dev(key, nonce, 0, aad if rank == 0 else None, total_aad_len, local_src, n, start, total_len,
                          local_dst, share)
    else:
        # each callable defaults on its own: only one injected -> the other is the HIP engine's
        cipher = cipher or _dev_cipher
        partial = partial or _dev_partial
        ctr0 = bytes(nonce) + b"\x00\x00\x00\x01"
        if n:
            cipher(key, ctr0, 1 + start // 16, local_src, local_dst, n)
        share = partial(key, nonce, aad if rank == 0 else None, total_aad_len, local_dst, n, start, total_len) \
            if takes_part else None
    return _xor_shares(_exchange(share, world, getattr(local_dst, "device", None), gather))


def gcm_decrypt_sharded(key, nonce, aad, total_aad_len, total_len, tag, local_src, local_dst, rank, world,
                        cipher=None, partial=None, gather=None):
    """Decrypt this rank's shard of ONE GCM message: 0, or 0x1A with local_dst untouched on EVERY rank (N7 across
    ranks, micro_aes.c:1200-1208).  Phase 1: every rank's share over its shard of the received ciphertext and the
    16-byte exchange; every rank compares the XOR with `tag` (16 bytes); phase 2, only on a match: the CTR pass."""
    cipher = cipher or _dev_cipher                               # each defaults on its own: a supplied one is kept
    partial = partial or _dev_partial
    start, n, takes_part = gcm_shard_roles(total_len, rank, world)
    share = partial(key, nonce, aad if rank == 0 else None, total_aad_len, local_src, n, start, total_len) \
        if takes_part else None
    got = _xor_shares(_exchange(share, world, getattr(local_src, "device", None), gather))
    diff = 0
    for a, b in zip(got, bytes(tag)):                            # no early exit
        diff |= a ^ b
    if diff or len(bytes(tag)) != 16:
        return 0x1A
    if n:
        cipher(key, bytes(nonce) + b"\x00\x00\x00\x01", 1 + start // 16, local_src, local_dst, n)
    return 0
