#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X AES engine.

Metric (BASELINE.json): GiB/s encrypted, AES-128-CTR over a 1 GiB synthetic
buffer per GPU (configs[1]); with --gpus N each rank encrypts its own 1 GiB
shard of an N GiB stream (configs[4], weak scaling, no data-path collective --
the optional ciphertext all-gather is timed separately and never part of
`value`).  A "step" is one full pass of the hot path over the resident buffer.

    python bench.py                       # 1 GPU, defaults
    python bench.py --gpus N              # launches its own N ranks (torch.distributed.run, free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against
the 8 TB/s HBM peak with ALGORITHMIC bytes (32 B per 16-byte block: read the
plaintext, write the ciphertext); `cpu_baseline` times the
compiled REFERENCE (oracle/_ref, or the oracle restatement if that did not
travel) on this box's host cores on a bounded sample of the same workload.
Other workloads (--workload ecb|xts|gcm|ocb|...) are for profiling, not the bench line; `cbc-enc` / `cmac` time ONE
serial chain with the reference's CPU loop beside it.

After the K timed steps the same step runs back to back for --sustain-s seconds (default 2): `sustained` and
`roofline.frac_sustained` ride next to `value`, never inside it.  With --gpus N the control plane is gloo; RCCL is
brought up beside it, proven with a 16-byte all-gather and used only if every rank saw it work -- otherwise the run
finishes on gloo and says so in `collective_backend` (setup_collectives).

With more than one rank the line also carries what SURVEY.md 8e lists beside the encrypt-only rate, all of it AFTER the
timed steps and never part of `value`: `gather_ms` (the ciphertext all-gather: RCCL over xGMI, or -- when RCCL did not
come up, e.g. the one-device dry run -- a gloo gather to rank 0), `encrypt_plus_gather_gib_s` (one step + the gather,
end to end), `gathered_stream_digest_ok` (the concatenated stream against the reference's digests: every 1 GiB shard,
and the whole 8 GiB C5 stream at N = 8) and `c_gather` (rank 0 alone runs the C host's uaes_mgpu_ctr_encrypt_gather
over all N devices in ONE process).  Each runs under its own bounded wait and try/except: none of them can cost the
line.  --no-gather / --no-c-gather switch them off.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_LINE_OUT = sys.stdout
GIB = 1 << 30
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
KEY16 = bytes(range(16))
KEY64 = bytes(range(64))
NONCE = bytes(range(0xF0, 0xFC))
CTR0 = NONCE + b"\x00\x00\x00\x01"   # iv || CTR_START_VALUE (micro_aes.c:968-971)


def splitmix_device(torch, seed, nbytes, word0, device):
    """SURVEY.md 8d synthetic stream, generated on the GPU (int64 wraps mod 2^64)."""
    out = torch.empty(nbytes // 8, dtype=torch.int64, device=device)
    chunk = 1 << 24

    def s64(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >> 63 else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    for o in range(0, out.numel(), chunk):
        n = min(chunk, out.numel() - o)
        w = torch.arange(word0 + o + 1, word0 + o + 1 + n, dtype=torch.int64, device=device)
        z = w * s64(0x9E3779B97F4A7C15) + s64(seed)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        out[o:o + n] = z ^ lsr(z, 31)
    return out.view(torch.uint8)


def sha_of(t):
    h = hashlib.sha256()
    step = 1 << 28
    for o in range(0, t.numel(), step):
        h.update(t[o:o + step].cpu().numpy().tobytes())
    return h.hexdigest()


def other_configs(torch, uaes, dev, st, steps=10, settle_ms=120.0):
    """BASELINE configs[3] and configs[2] beside the headline, AFTER its timed steps and never inside `value`
    (VERDICT r05 next #2): AES-128-GCM over 1 GiB (seed 4; AES_GCM_encrypt, micro_aes.c:1164-1179) and AES-256-XTS over
    2^20 sectors of 4 KiB (seed 3; AES_XTS_encrypt per sector, micro_aes.c:1066-1093), each settled for `settle_ms`, then
    `steps` steps timed with HIP events on the launch stream, then checked against the compiled reference's results
    (tests/golden/digests.json: the C4 tag; SHA-256 of the whole 4 GiB C3 ciphertext).  frac = 32 B per 16-byte block
    over the mean step time against the 8 TB/s peak, as in `roofline`."""
    with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
        gold = json.load(f)

    def timed(step, n):
        step()
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        while (time.perf_counter() - w0) * 1e3 < settle_ms:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps):
            step()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        return {"gib_s": round(n / GIB / (ms * 1e-3), 1), "ms_per_step": round(ms, 4), "steps": steps,
                "frac": round(2.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes": n}

    out = {}
    try:
        n = GIB
        src = splitmix_device(torch, 4, n, 0, dev)
        dst = torch.empty(n + 16, dtype=torch.uint8, device=dev)
        r = timed(lambda: uaes.gcm_encrypt_dev(KEY16, NONCE, None, src, n, dst, stream=st), n)
        tag = bytes(dst[n:].cpu().numpy()).hex()
        r["workload"] = "AES-128-GCM, 1 GiB + tag, seed 4 (configs[3])"
        r["verified"] = tag == gold["C4_gcm128_1GiB_seed4"]["tag"]
        r["check"] = "tag %s against the reference's C4 tag" % tag
        out["gcm_c4"] = r
        del src, dst
    except Exception as e:                                          # noqa: BLE001 -- never costs the headline
        out["gcm_c4"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    try:
        n = 4 * GIB
        src = splitmix_device(torch, 3, n, 0, dev)
        dst = torch.empty(n, dtype=torch.uint8, device=dev)
        r = timed(lambda: uaes.xts_sectors_dev(KEY64, 0, 4096, n // 4096, src, dst, stream=st), n)
        r["workload"] = "AES-256-XTS, 2^20 sectors x 4 KiB, seed 3 (configs[2])"
        r["verified"] = sha_of(dst) == gold["C3_xts256_2p20_sectors_seed3"]["sha256"]
        r["check"] = "SHA-256 of the 4 GiB ciphertext against the reference's C3 digest"
        out["xts_c3"] = r
        del src, dst
    except Exception as e:                                          # noqa: BLE001
        out["xts_c3"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    torch.cuda.empty_cache()
    return out


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read())))
            break
        except Exception:
            continue
    return max(1, min(n, 128))


def measure_counters(workload, nbytes, counters=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")):
    """PMC counters per step, measured live: one SEPARATE rocprofv3 pass per counter (kernel trace only, as
    MI355X_MICROARCH.md prescribes) over a short child run of this very command (3 steps + 1 warm-up, no settling, no
    CPU leg).  Per step = the counter of every kernel of the engine (names k_*) summed over the child's launches / its
    4 steps.  Returns ({counter: value per step}, None) or ({whatever was collected}, why the rest is missing)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {}, "rocprofv3 not found"
    steps, warm = 3, 1
    got = {}
    tmp = tempfile.mkdtemp(prefix="uaes_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for ctr in counters:
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--bytes", str(nbytes),
                   "--steps", str(steps), "--warmup", str(warm), "--settle-ms", "0", "--no-cpu", "--no-verify",
                   "--no-traffic", "--no-clock-probe", "--sustain-s", "0", "--no-c-gather", "--no-other-configs"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return got, "rocprofv3 --pmc %s pass failed (rc %d)" % (ctr, r.returncode)
            total = 0.0
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row.get("Kernel_Name", "")
                        if row.get("Counter_Name") == ctr and (name.startswith("k_") or name.startswith("void k_")):
                            total += float(row.get("Counter_Value", 0))
            got[ctr] = total / (steps + warm)
    except Exception as e:                      # a profiler problem must not cost the bench line
        return got, "PMC pass: %s" % e
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got, None


def traffic_of(counters):
    """HBM bytes per step: FETCH_SIZE and WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE reports half of a wide coalesced
    read stream (MI355X_MICROARCH.md, HBM), so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  (bytes, note) or (None, None)."""
    if "FETCH_SIZE" not in counters or "WRITE_SIZE" not in counters:
        return None, None
    traffic = int((2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024)
    return traffic, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE as two separate "
                     "passes over a 4-step child run; (2 x %.0f + %.0f) KiB per step" % (counters["FETCH_SIZE"], counters["WRITE_SIZE"]))


def valu_per_block_of(counters, nbytes, workload, profiles_dir=None):
    """VALU instructions per 16-byte block = SQ_INSTS_VALU (wave instructions per step) x 64 lanes / blocks per step:
    from the PMC pass of THIS run when there is one, else the figure tools/profile.sh recorded for this workload and
    size in profiles/pmc_valu.json -- and then labelled as recorded, with the build it came from.  (value, source)"""
    if counters.get("SQ_INSTS_VALU"):
        return round(counters["SQ_INSTS_VALU"] * 64.0 / (nbytes / 16.0), 1), \
            "measured in this run: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU over a 4-step child run"
    try:
        with open(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "pmc_valu.json")) as f:
            t = json.load(f).get(workload)
        if t and t["bytes_per_gpu"] == nbytes:
            return t["valu_insts_per_block"], "recorded in %s (build %s; not measured in this run)" % (t["source"], t.get("build", "?"))
    except Exception:
        pass
    return None, None


def cpu_baseline(workload, sample=None, all_cores=True):
    """Reference CPU path on this host: 1 core (the reference is single-threaded
    and non re-entrant, micro_aes.c:72) and, via fork, all cores.  `sample` (bytes) and `all_cores=False` shrink it
    for the tests; the bench line uses the defaults."""
    import numpy as np
    from oracle.pyoracle import Oracle, Reference
    orc = Oracle()
    use_ref = Reference.available(128)
    sample = sample or {"ctr": 512 << 20, "ecb": 512 << 20, "xts": 512 << 20, "gcm": 64 << 20, "cbc-enc": 256 << 20,
                        "cmac": 256 << 20}.get(workload, 256 << 20)
    mac16 = (ctypes.c_uint8 * 16)()
    buf = np.empty(sample, dtype=np.uint8)
    orc.splitmix_into(2, buf)
    out = np.empty(sample + 16, dtype=np.uint8)
    src, dst = ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(out.ctypes.data)
    if use_ref:
        L = Reference(256 if workload == "xts" else 128).L
        fn = {"ctr": lambda n: L.AES_CTR_encrypt(KEY16, NONCE, src, n, dst),
              "ecb": lambda n: L.AES_ECB_encrypt(KEY16, src, n, dst),
              "xts": lambda n: [L.AES_XTS_encrypt(KEY64, (s).to_bytes(16, "little"),
                                                  ctypes.c_void_p(buf.ctypes.data + s * 4096), 4096,
                                                  ctypes.c_void_p(out.ctypes.data + s * 4096))
                                for s in range(n // 4096)],
              "gcm": lambda n: L.AES_GCM_encrypt(KEY16, NONCE, None, 0, src, n, dst),
              "ocb": lambda n: L.AES_OCB_encrypt(KEY16, NONCE, None, 0, src, n, dst),
              "ocb-dec": lambda n: L.AES_OCB_encrypt(KEY16, NONCE, None, 0, src, n, dst),
              "cbc-dec": lambda n: L.AES_CBC_decrypt(KEY16, bytes(range(16)), src, n, dst),
              "cbc-enc": lambda n: L.AES_CBC_encrypt(KEY16, bytes(range(16)), src, n, dst),
              "cmac": lambda n: L.AES_CMAC(KEY16, src, n, mac16),
              "cfb-dec": lambda n: L.AES_CFB_decrypt(KEY16, bytes(range(16)), src, n, dst)}[workload]
    else:
        L = orc.L
        fn = {"ctr": lambda n: L.orc_ctr_encrypt(128, KEY16, NONCE, src, n, dst),
              "ecb": lambda n: L.orc_ecb_encrypt(128, KEY16, src, n, dst),
              "xts": lambda n: L.orc_xts_sectors(256, KEY64, 0, 4096, n // 4096, src, dst, 1),
              "gcm": lambda n: L.orc_gcm_encrypt(128, KEY16, NONCE, None, 0, src, n, dst),
              "ocb": lambda n: L.orc_ocb_encrypt(128, KEY16, NONCE, None, 0, src, n, dst),
              "ocb-dec": lambda n: L.orc_ocb_encrypt(128, KEY16, NONCE, None, 0, src, n, dst),
              "cbc-dec": lambda n: L.orc_cbc_decrypt(128, KEY16, bytes(range(16)), src, n, dst),
              "cbc-enc": lambda n: L.orc_cbc_encrypt(128, KEY16, bytes(range(16)), src, n, dst),
              "cmac": lambda n: L.orc_cmac(128, KEY16, src, n, mac16),
              "cfb-dec": lambda n: L.orc_cfb(128, KEY16, bytes(range(16)), 0, src, n, dst)}[workload]
    t0 = time.perf_counter()
    fn(sample)
    t1 = time.perf_counter() - t0
    res = {"value": round(sample / GIB / t1, 5), "unit": "GiB/s", "cores": 1,
           "kind": "reference" if use_ref else "port",
           "sample": "%d MiB of the same synthetic %s workload, single call, gcc -O3" % (sample >> 20, workload)}
    if not all_cores:
        return res
    # all cores: fork P processes over contiguous shards (throughput-equivalent;
    # the reference API cannot start a CTR shard at an offset)
    P = usable_cores()
    per = max((sample // 16) // 4096 * 4096, 4096)
    t0 = time.perf_counter()
    pids = []
    for _ in range(P):
        pid = os.fork()
        if pid == 0:
            fn(per)
            os._exit(0)
        pids.append(pid)
    for pid in pids:
        os.waitpid(pid, 0)
    tp = time.perf_counter() - t0
    res["all_cores"] = {"value": round(P * per / GIB / tp, 4), "unit": "GiB/s", "cores": P,
                        "sample": "%d forked processes x %d MiB" % (P, per >> 20)}
    try:
        with open("/proc/cpuinfo") as f:
            res["cpu"] = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    return res


def setup_collectives(a, torch, rank, world, local, dev):
    """Control plane of an N-rank run.  `value` needs no data-path collective: a barrier either
    side of the timed steps and three scalar reductions.  The default process group is therefore gloo over
    127.0.0.1 -- it cannot fail for reasons that have to do with the GPUs -- and RCCL is brought up NEXT to it
    as a second group, proven with a 16-byte all-gather under a bounded wait, and used for the barrier, the
    reductions and the optional ciphertext gather only once EVERY rank has seen it work.  If any rank's RCCL
    init or probe raises or does not return in time, all ranks agree (over gloo) to stay on gloo: every rank
    still runs on its own GPU and still verifies its shard, and the line says so in `collective_backend`.
    Returns (dist, group, device for collective scalars, info for the JSON line)."""
    import datetime
    import threading
    import torch.distributed as dist
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")      # a stuck RCCL probe must not abort the process
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
    info = {"collective_backend": "gloo", "control_plane": "gloo over 127.0.0.1"}
    if a.backend != "nccl":
        info["collective_backend"] = "gloo (requested with --backend gloo)"
        return dist, None, torch.device("cpu"), info
    wait_s = float(os.environ.get("UAES_BENCH_RCCL_WAIT_S", "90"))
    state = {"ok": False, "err": "no answer within %.0f s" % wait_s, "group": None}

    def bring_up():
        try:
            if os.environ.get("UAES_BENCH_FORCE_NCCL_FAIL"):
                raise RuntimeError("forced by UAES_BENCH_FORCE_NCCL_FAIL")
            torch.cuda.set_device(local)
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(wait_s * 4, 600)))
            state["group"] = g
            mine = torch.full((16,), rank, dtype=torch.uint8, device=dev)
            allr = torch.empty(16 * world, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(allr, mine, group=g)
            torch.cuda.synchronize()
            want = torch.arange(world, dtype=torch.uint8).repeat_interleave(16)
            if not torch.equal(allr.cpu(), want):
                raise RuntimeError("16-byte all-gather returned the wrong bytes")
            state["ok"], state["err"] = True, None
        except Exception as e:                                      # noqa: BLE001 -- whatever RCCL throws
            state["err"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")

    th = threading.Thread(target=bring_up, daemon=True)
    th.start()
    th.join(wait_s)
    flag = torch.tensor([1 if state["ok"] and not th.is_alive() else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                      # gloo: every rank learns whether ALL ranks have RCCL
    if int(flag.item()) == 1:
        info = {"collective_backend": "nccl (RCCL)", "rccl_ranks": dist.get_world_size(state["group"]),
                "rccl_probe": "16-byte all_gather_into_tensor ok on every rank", "control_plane": "gloo init, RCCL barrier/reductions"}
        return dist, state["group"], dev, info
    why = state["err"] if not state["ok"] else "another rank's RCCL bring-up failed"
    info["collective_backend"] = "gloo (nccl init failed: %s)" % why
    info["rccl_stuck_thread"] = th.is_alive()
    return dist, None, torch.device("cpu"), info


def resolve_gather_flags(a, world, force=False):
    """--gather / --c-gather default ON as soon as there is more than one rank (SURVEY.md 8e "what to report":
    encrypt-only, encrypt + gather, bit-exactness of the concatenated stream) or --force-collective asks for the
    one-rank run of the same code, OFF for the plain one-GPU line"""
    many = world > 1 or force
    gather = many if a.gather is None else bool(a.gather)
    c_gather = (many and gather) if a.c_gather is None else bool(a.c_gather)
    return gather, c_gather and gather


def bounded(fn, wait_s):
    """fn() in a daemon thread, waited for at most wait_s seconds: (result, error text or None, thread still alive).
    Whatever a collective or the C host throws -- or if it never returns -- the caller gets an answer."""
    import threading
    state = {"res": None, "err": "no answer within %.0f s" % wait_s}

    def run():
        try:
            state["res"] = fn()
            state["err"] = None
        except BaseException as e:                                  # noqa: BLE001
            msg = str(e).splitlines()[0][:300] if str(e) else ""
            state["err"] = "%s: %s" % (type(e).__name__, msg)

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(wait_s)
    alive = th.is_alive()
    return (None if alive else state["res"]), (state["err"] if (alive or state["err"]) else None), alive


def check_gathered_stream(torch, full, world, n, workload, seed, no_verify):
    """rank 0: the concatenated ciphertext against the reference.  CTR at 1 GiB per rank: SHA-256 of every shard
    against the reference's digest of that shard of the C5 stream and, at N = 8, of the whole 8 GiB stream
    (BASELINE configs[4], tests/golden/digests.json).  Other sizes / workloads: the head of every shard against the
    oracle at that shard's offset.  Returns a dict of fields for the line."""
    if no_verify:
        return {}
    if workload == "ctr" and n == GIB and world <= 8:
        with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
            gold = json.load(f)
        whole, ok, step = hashlib.sha256(), True, 1 << 28
        for g in range(world):
            h = hashlib.sha256()
            for o in range(g * n, (g + 1) * n, step):
                b = full[o:o + step].cpu().numpy().tobytes()
                h.update(b)
                whole.update(b)
            ok = ok and h.hexdigest() == gold["C5_shard_%d" % g]["sha256"]
        res = {"gathered_stream_digest_ok": ok,
               "gathered_stream_check": "SHA-256 of each of the %d gathered 1 GiB shards against the reference's digest" % world}
        if world == 8:
            res["gathered_stream_digest_ok"] = ok and whole.hexdigest() == gold["C5_ctr128_8GiB_seed2"]["sha256"]
            res["gathered_stream_check"] += " and of the whole 8 GiB stream against C5_ctr128_8GiB_seed2"
        return res
    if workload in ("ctr", "ecb", "xts"):
        from oracle.pyoracle import Oracle
        orc = Oracle()
        m, ok = min(n, 1 << 16), True
        for g in range(world):
            head = orc.splitmix(seed, m, word0=g * (n // 8))
            got = bytes(full[g * n:g * n + m].cpu().numpy())
            if workload == "ctr":
                ok = ok and got == orc.ctr_xcrypt_at(KEY16, CTR0, g * (n // 16), head)
            elif workload == "ecb":
                ok = ok and got == orc.ecb_encrypt(KEY16, head)
            else:
                ok = ok and got == orc.xts_sectors(KEY64, g * (n // 4096), 4096, head, True)[1]
        return {"gathered_stream_digest_ok": ok,
                "gathered_stream_check": "head (%d KiB) of each of the %d gathered shards against the oracle" % (m >> 10, world)}
    return {}


def gather_phase(a, torch, dist, cg, rank, world, local, n, dev, dst, step, seed):
    """After the timed steps: the ciphertext gather, one encrypt + gather end to end, and the check of the
    concatenated stream.  RCCL (all_gather_into_tensor over xGMI, every rank receives the stream) when it is up; else
    a gloo gather to rank 0 through host memory in 64 MiB pieces -- the one-device dry run's stand-in, labelled as such.
    Returns (fields for the line, RCCL still usable)."""
    wait_s = float(os.environ.get("UAES_BENCH_GATHER_WAIT_S", "300"))
    rccl = cg is not None
    # the stand-in gather runs in a thread: it gets a gloo group of its own, so that the control plane's collectives on
    # the default group (main thread) can never interleave with a gather that is late or stuck
    gg = None if rccl else dist.new_group(backend="gloo")

    def run():
        torch.cuda.set_device(local)
        if rccl:
            full = torch.empty(world * n, dtype=torch.uint8, device=dev)

            def gather():
                dist.all_gather_into_tensor(full, dst[:n], group=cg)
                torch.cuda.synchronize()
                dist.barrier(group=cg)
        else:
            chunk = min(n, 64 << 20)
            full = torch.empty(world * n, dtype=torch.uint8, device=dev) if rank == 0 else None
            host = torch.empty(chunk, dtype=torch.uint8).pin_memory()
            recv = [torch.empty(chunk, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None

            def gather():
                for o in range(0, n, chunk):
                    m = min(chunk, n - o)
                    host[:m].copy_(dst[o:o + m])
                    dist.gather(host[:m], gather_list=[r[:m] for r in recv] if rank == 0 else None, dst=0, group=gg)
                    if rank == 0:
                        for g in range(world):
                            full[g * n + o:g * n + o + m].copy_(recv[g][:m])
                torch.cuda.synchronize()
                dist.barrier(group=gg)
        gather()                                                 # first use sets the channels up
        g0 = time.perf_counter()
        gather()
        gather_ms = (time.perf_counter() - g0) * 1e3
        e0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        gather()
        e2e_ms = (time.perf_counter() - e0) * 1e3
        return full, gather_ms, e2e_ms

    res, err, stuck = bounded(run, wait_s)
    flag = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                  # gloo: did EVERY rank get through?
    out = {}
    if int(flag.item()) != 1:
        out["gather"] = {"error": err or "another rank's gather failed", "stuck_thread": stuck}
        # RCCL (if that is what carried it) is no longer trusted by ANY rank; tearing its group down could hang with
        # a peer's stuck thread, so every rank leaves through os._exit once the line is out
        return out, False, stuck or rccl
    full, gather_ms, e2e_ms = res
    out["gather_ms"] = round(gather_ms, 3)
    out["gather_backend"] = ("RCCL all_gather_into_tensor, ONE rank (--force-collective: the code path, not xGMI)" if rccl and world == 1 else
                             "RCCL all_gather_into_tensor over xGMI (every rank receives the whole stream)" if rccl else
                             "gloo gather to rank 0 through host memory (RCCL not up: dry run, says nothing about xGMI)")
    out["gather_gib_s"] = round(world * n / GIB / (gather_ms * 1e-3), 2)
    out["encrypt_plus_gather_ms"] = round(e2e_ms, 3)
    out["encrypt_plus_gather_gib_s"] = round(world * n / GIB / (e2e_ms * 1e-3), 2)
    if rank == 0:
        try:
            out.update(check_gathered_stream(torch, full, world, n, a.workload, seed, a.no_verify))
        except Exception as e:                                   # noqa: BLE001
            out["gathered_stream_check"] = "failed: %s: %s" % (type(e).__name__, str(e)[:200])
    del full
    dist.barrier()                                               # gloo: the others wait for rank 0's hashing
    return out, rccl, False


def c_gather_phase(a, torch, uaes, rank, world, n, dev, src, dst, seed):
    """rank 0 alone: the C host's own encrypt + gather (north_star: host code in C, RCCL only for the final ciphertext
    gather) -- one process over all N devices, fresh shards generated on every device, timed end to end, twice."""
    ndev = 1 if a.single_device else world
    devs = [0] * world if a.single_device else list(range(world))
    ins, outs, keep = (ctypes.c_void_p * world)(), (ctypes.c_void_p * world)(), []
    for g, d in enumerate(devs):
        dd = torch.device("cuda", d)
        t_in = src if g == 0 else splitmix_device(torch, seed, n, g * (n // 8), dd)
        t_out = torch.empty(n, dtype=torch.uint8, device=dd)
        keep += [t_in, t_out]
        ins[g], outs[g] = t_in.data_ptr(), t_out.data_ptr()
    full = torch.empty(world * n, dtype=torch.uint8, device=dev)
    for d in set(devs):
        torch.cuda.synchronize(d)
    times = []
    stats0 = (ctypes.c_ulong * 5)()
    uaes.engine().uaes_debug_gather_stats(stats0)
    for _ in range(2):
        g0 = time.perf_counter()
        rc = uaes.engine().uaes_mgpu_ctr_encrypt_gather(world, (ctypes.c_int * world)(*devs), 128, KEY16, CTR0, 0,
                                                        ins, world * n, outs, 0, ctypes.c_void_p(full.data_ptr()))
        for d in set(devs):
            torch.cuda.synchronize(d)
        times.append((time.perf_counter() - g0) * 1e3)
        if rc != 0:
            raise RuntimeError(uaes.engine().uaes_last_error().decode())
    stats1 = (ctypes.c_ulong * 5)()
    uaes.engine().uaes_debug_gather_stats(stats1)
    res = {"ms_first": round(times[0], 3), "ms": round(times[1], 3), "devices": ndev,
           "gib_s_end_to_end": round(world * n / GIB / (times[1] * 1e-3), 1),
           "call": "uaes_mgpu_ctr_encrypt_gather (C host, RCCL send/recv to device 0)",
           # what RCCL itself was asked to do by the two calls (uaes_debug_gather_stats): 0 sends = every slice was local
           "rccl_sends": int(stats1[0] - stats0[0]), "rccl_recvs": int(stats1[1] - stats0[1]),
           "rccl_comm_inits": int(stats1[3] - stats0[3]),
           "forced_self_send": bool(os.environ.get("UAES_GATHER_FORCE_RCCL"))}
    if not a.no_verify:
        chk = check_gathered_stream(torch, full, world, n, "ctr", seed, False)
        if "gathered_stream_digest_ok" in chk:
            res["stream_digest_ok"] = chk["gathered_stream_digest_ok"]
            res["stream_check"] = chk["gathered_stream_check"]
        res["shard0_equals_own_step"] = bool(torch.equal(full[:n], dst[:n]))
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU, exactly as
    the documented command does (torch.distributed.run, rendezvous on 127.0.0.1, a free port), relay their
    output and exit with their status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ctr", choices=["ctr", "ecb", "xts", "gcm", "ocb", "ocb-dec", "cbc-dec", "cfb-dec", "cbc-enc", "cmac"])
    ap.add_argument("--bytes", type=int, default=GIB, help="bytes per GPU")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="extra untimed warm-up until the clocks have settled (0 = only --warmup steps)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the shader-clock sample behind roofline.lds_ceiling (taken in the middle of the sustained run)")
    ap.add_argument("--sustain-s", type=float, default=2.0,
                    help="after the K timed steps, seconds of the same step back to back (untimed for `value`) behind "
                         "`sustained` and roofline.frac_sustained; 0 = skip")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="after the timed steps: the ciphertext gather, encrypt + gather end to end and the digest of the "
                         "concatenated stream (default: ON with more than one rank)")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    ap.add_argument("--c-gather", dest="c_gather", action="store_true", default=None,
                    help="with the gather: rank 0 also runs uaes_mgpu_ctr_encrypt_gather -- the C host's own encrypt + RCCL "
                         "gather over all N devices in ONE process (include/uaes_hip.h) -- while the other ranks wait "
                         "(default: ON with more than one rank)")
    ap.add_argument("--no-c-gather", dest="c_gather", action="store_false")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --single-device: dry-run the multi-rank code path on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="all ranks use cuda:0 (dry run only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip `other_configs` (GCM C4 and XTS C3 timed and verified after the headline; 1-GPU CTR 1 GiB line only)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: bring the collectives up anyway -- a ONE-rank RCCL process group "
                         "(init_process_group / new_group(\"nccl\"), world_size 1) carries the barrier, the reductions and "
                         "the ciphertext all-gather, and the C host's uaes_mgpu_ctr_encrypt_gather sends its slice through "
                         "ncclSend / ncclRecv to itself (UAES_GATHER_FORCE_RCCL) -- so that a one-GPU box executes the "
                         "code an 8-GPU run depends on.  Says nothing about xGMI; never part of `value`.")
    return ap


def main():
    a = build_parser().parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)

    # ONE JSON line on stdout and nothing else: libraries below us write to the C-level stdout (RCCL prints its version
    # banner there when a communicator is first made), so file descriptor 1 is pointed at stderr for the life of the
    # process and the line goes out through a private duplicate of the real stdout.
    global _LINE_OUT
    sys.stdout.flush()
    _LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import micro_aes_amd as uaes

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("--gpus %d but the launcher started %d rank(s)" % (a.gpus, world))
    if a.single_device:
        local = 0
    elif local >= torch.cuda.device_count():
        sys.exit("rank %d wants cuda:%d but only %d device(s) are visible (--single-device is the 1-GPU dry run)"
                 % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist, cg, cdev, coll_info = None, None, dev, {}
    force = bool(a.force_collective) and world == 1
    if force:
        # a one-rank job has no launcher: give torch.distributed its rendezvous (127.0.0.1, a free port)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port))):
            os.environ.setdefault(k, v)
        os.environ["UAES_GATHER_FORCE_RCCL"] = "1"          # read by the C host on every gather call
    if world > 1 or force:
        dist, cg, cdev, coll_info = setup_collectives(a, torch, rank, world, local, dev)
        if force:
            coll_info["force_collective"] = "one-rank groups: every collective below ran through RCCL on this one GPU"
    rccl_up = cg is not None                                     # RCCL over xGMI carries the collectives

    n = a.bytes
    seed = {"ctr": 2, "ecb": 1, "xts": 3, "gcm": 4, "cbc-dec": 5, "cfb-dec": 6, "ocb": 7, "ocb-dec": 7, "cbc-enc": 5, "cmac": 5}[a.workload]
    # rank g owns bytes [g*n, (g+1)*n) of the world*n stream
    src = splitmix_device(torch, seed, n, rank * (n // 8), dev)
    dst = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()

    if a.workload == "ctr":
        def step():
            uaes.ctr_xcrypt_dev(KEY16, CTR0, rank * (n // 16), src, dst, nbytes=n, stream=st)
    elif a.workload == "ecb":
        def step():
            uaes.ecb_dev(KEY16, src, dst, nbytes=n, stream=st)
    elif a.workload == "xts":
        def step():
            uaes.xts_sectors_dev(KEY64, rank * (n // 4096), 4096, n // 4096, src, dst, stream=st)
    elif a.workload == "gcm":
        def step():
            uaes.gcm_encrypt_dev(KEY16, NONCE, None, src, n, dst, stream=st)
    elif a.workload == "ocb":
        def step():
            uaes.ocb_dev(KEY16, NONCE, None, src, n, dst, stream=st)
    elif a.workload == "ocb-dec":
        ocb_ct = torch.empty(n + 16, dtype=torch.uint8, device=dev)
        ocb_status = torch.full((1,), -1, dtype=torch.int32, device=dev)
        uaes.ocb_dev(KEY16, NONCE, None, src, n, ocb_ct, stream=st)

        def step():
            uaes.ocb_dev(KEY16, NONCE, None, ocb_ct, n, dst, decrypt=True, status=ocb_status, stream=st)
    elif a.workload in ("cbc-enc", "cmac"):
        # ONE serial chain (north_star: "CBC/CFB/OFB stay single-GPU because the chain is serial"): a latency-bound
        # single wave, here so that the reference's CPU loop is timed beside it in the same line (use --bytes 4194304)
        import ctypes as C
        L = uaes.engine()
        iv16, mac16 = bytes(range(16)), (C.c_uint8 * 16)()

        def step():
            if a.workload == "cbc-enc":
                assert L.uaes_cbc_encrypt(128, KEY16, iv16, C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())) == 0
            else:
                assert L.uaes_cmac(128, KEY16, C.c_void_p(src.data_ptr()), n, mac16) == 0
    else:
        # block-parallel decrypt directions of the feedback modes, through the host-pointer C ABI
        # with device pointers (synchronous call: launch + stream sync)
        import ctypes as C
        L = uaes.engine()
        fn = L.uaes_cbc_decrypt if a.workload == "cbc-dec" else L.uaes_cfb_decrypt
        iv16 = bytes(range(16))

        def step():
            assert fn(128, KEY16, iv16, C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())) == 0

    probe_out, side = None, None
    if not a.no_clock_probe and a.sustain_s > 0:
        try:
            probe_out = torch.zeros(2, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream(device=dev)
        except Exception:
            probe_out = None
    for _ in range(a.warmup):
        step()
    # The GPU's clocks take tens of milliseconds of load to settle (a 1 GiB step is ~0.7 ms:
    # three of them end before the ramp does and the next twenty measure it).  Keep warming
    # up, untimed, until `--settle-ms` of work has been queued; reported as warmup_extra_steps.
    extra = 0
    if a.settle_ms > 0:
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        while (time.perf_counter() - w0) * 1e3 < a.settle_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            extra += 8
    torch.cuda.synchronize()
    if dist:
        dist.barrier(group=cg)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(st)
    for i in range(a.steps):
        step()
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    if dist:
        dist.barrier(group=cg)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
    # Untimed for `value`, after the measured steps: the SUSTAINED rate.  K steps after 150 ms of settling are a
    # window of ~12 ms; the power manager keeps adjusting for longer than that (1605 GiB/s in the window,
    # 1512-1524 over 30 000 steps).  So the same step is queued back to back for >= --sustain-s seconds (HIP events on
    # the launch stream around the whole run) and, in the middle of it, a one-wave probe on a second stream counts shader
    # cycles against the constant 100 MHz counter for 20 ms: the clock the chip really runs at under this load (the
    # device properties quote 2.4 GHz; at the 1.4 kW cap a cipher kernel settles near 2.0 GHz).
    sclk_mhz, sustained = None, None
    if a.sustain_s > 0:
        mean_ms = max(sum(per_step_ms) / len(per_step_ms), 1e-3)
        sus_steps = max(a.steps, int(a.sustain_s * 1e3 / mean_ms) + 1)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        probe_rc = -1
        s0.record(st)
        for i in range(sus_steps):
            step()
            if i == sus_steps // 2 and probe_out is not None:
                probe_rc = uaes.engine().uaes_clock_probe_dev(ctypes.c_void_p(probe_out.data_ptr()), 20000,
                                                              ctypes.c_void_p(side.cuda_stream))
        s1.record(st)
        torch.cuda.synchronize()
        sus_ms = s0.elapsed_time(s1)
        if probe_out is not None and probe_rc == 0:
            cyc, ticks = [int(x) for x in probe_out.tolist()]
            if ticks > 0:
                sclk_mhz = cyc / (ticks / 100.0)
        sustained = {"gib_s": n * sus_steps / GIB / (sus_ms * 1e-3), "seconds": sus_ms * 1e-3, "steps": sus_steps,
                     "ms_per_step": sus_ms / sus_steps}
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=cg)
        elapsed = float(t.item())

    # ---- beside the encrypt-only rate: gather, encrypt + gather, the concatenated stream, the C host's
    # own gather.  After the timed region, each under a bounded wait; a failure becomes a field of the line.
    want_gather, want_c_gather = resolve_gather_flags(a, world, force)
    gather_info, c_gather, stuck_extra = {}, None, False
    if dist and want_gather:
        try:
            gather_info, still_rccl, stuck_extra = gather_phase(a, torch, dist, cg, rank, world, local, n, dev, dst, step, seed)
        except Exception as e:                                  # noqa: BLE001 -- the bench line must survive
            gather_info, still_rccl = {"gather": {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}}, False
        if cg is not None and not still_rccl and "gather" in gather_info:
            cg, cdev = None, torch.device("cpu")                # RCCL misbehaved: the remaining reductions go over gloo
            coll_info["collective_backend"] += "; gloo after the gather failed"
    rccl_self = (world == 1 and not force and a.workload == "ctr" and a.c_gather is not False and a.gather is not False)
    if rccl_self:
        # The plain one-GPU line: no process group at all, but the C host's gather still runs once through RCCL
        # (ncclSend / ncclRecv to itself, UAES_GATHER_FORCE_RCCL) so that every driver run executes the code that
        # configs[4] depends on and checks its stream against the reference.  After the timed steps; bounded.
        os.environ["UAES_GATHER_FORCE_RCCL"] = "1"
        res, err, stuck = bounded(lambda: c_gather_phase(a, torch, uaes, rank, world, n, dev, src, dst, seed),
                                  float(os.environ.get("UAES_BENCH_CGATHER_WAIT_S", "120")))
        os.environ.pop("UAES_GATHER_FORCE_RCCL", None)
        c_gather = res if err is None else {"error": err, "stuck_thread": stuck}
        if stuck:
            stuck_extra = True
    if want_gather and want_c_gather and a.workload == "ctr" and (world > 1 or force):
        if rank == 0:
            res, err, stuck = bounded(lambda: c_gather_phase(a, torch, uaes, rank, world, n, dev, src, dst, seed),
                                      float(os.environ.get("UAES_BENCH_CGATHER_WAIT_S", "300")))
            c_gather = res if err is None else {"error": err, "stuck_thread": stuck}
            stuck_extra = stuck_extra or stuck
        if dist:
            dist.barrier()                                      # gloo (the default group)

    # measured HBM stream-copy ceiling on this GPU (read n + write n), for context next to the 8 TB/s spec
    copy_gbs = None
    if world == 1:
        tmp = torch.empty(n, dtype=torch.uint8, device=dev)
        for _ in range(3):
            tmp.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(st)
        for _ in range(10):
            tmp.copy_(src)
        c1.record(st)
        torch.cuda.synchronize()
        copy_gbs = 2.0 * n * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del tmp

    verify = None
    if not a.no_verify:
        with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
            gold = json.load(f)
        if a.workload == "ctr" and n == GIB and rank < 8:
            # every rank checks its WHOLE shard against the reference's digest of that shard of the 8 GiB
            # stream (PRESET_COUNTER build, tests/golden/make_fixtures.py --big); shard 0 = C2
            verify = sha_of(dst[:n]) == gold["C5_shard_%d" % rank]["sha256"]
            assert gold["C5_shard_0"]["sha256"] == gold["C2_ctr128_1GiB_seed2"]["sha256"]
        elif a.workload == "gcm" and n == GIB and rank == 0:
            verify = bytes(dst[n:].cpu().numpy()).hex() == gold["C4_gcm128_1GiB_seed4"]["tag"]
        elif a.workload == "xts" and n == 4 * GIB and rank == 0:
            verify = sha_of(dst[:n]) == gold["C3_xts256_2p20_sectors_seed3"]["sha256"]
        else:
            # other ranks / sizes: head of the shard against the CPU oracle
            from oracle.pyoracle import Oracle
            orc = Oracle()
            m = min(n, 1 << 16)
            head = orc.splitmix(seed, m, word0=rank * (n // 8))
            got = bytes(dst[:m].cpu().numpy())
            if a.workload == "ctr":
                verify = got == orc.ctr_xcrypt_at(KEY16, CTR0, rank * (n // 16), head)
            elif a.workload == "ecb":
                verify = got == orc.ecb_encrypt(KEY16, head)
            elif a.workload == "xts":
                verify = got == orc.xts_sectors(KEY64, rank * (n // 4096), 4096, head, True)[1]
            elif a.workload == "ocb":
                # the head of an OCB ciphertext does not depend on what follows it
                verify = got[: m - 16] == orc.ocb_encrypt(KEY16, NONCE, b"", head)[: m - 16]
            elif a.workload == "ocb-dec":
                verify = got == head and int(ocb_status.item()) == 0
            elif a.workload == "cbc-dec":
                verify = got[: m - 32] == orc.cbc(KEY16, bytes(range(16)), head, False)[1][: m - 32]
            elif a.workload == "cfb-dec":
                verify = got == orc.cfb(KEY16, bytes(range(16)), head, False)
            elif a.workload == "cbc-enc":
                # CS3 stealing swaps the LAST two blocks of a message: the head of a longer message matches up to there
                verify = got[: m - 32] == orc.cbc(KEY16, bytes(range(16)), head, True)[1][: m - 32]
            elif a.workload == "cmac":
                verify = bytes(mac16) == orc.cmac(KEY16, orc.splitmix(seed, n)) if n <= (16 << 20) else None
        if dist:
            v = torch.tensor([1 if verify in (True, None) else 0], device=cdev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=cg)
            verify = bool(v.item())

    kern_ms = sum(per_step_ms) / len(per_step_ms)
    kern_all = [kern_ms]
    sus_all = [sustained["gib_s"]] if sustained else None
    if dist:
        t = torch.zeros(2 * world, dtype=torch.float64, device=cdev)
        t[rank] = kern_ms
        t[world + rank] = sustained["gib_s"] if sustained else 0.0
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=cg)
        kern_all = [float(x) for x in t.tolist()[:world]]
        if sustained:
            sus_all = [float(x) for x in t.tolist()[world:]]

    if rank == 0:
        total_gib = world * n * a.steps / GIB
        algo_bytes = 2.0 * n                       # read n + write n per launch, per GPU
        per_gpu = [algo_bytes / (k * 1e-3) / 1e9 for k in kern_all]
        achieved = per_gpu[0]
        # HBM bytes per step: measured live by two rocprofv3 PMC passes over a short child run (N = 1 only, like the
        # CPU leg); where that is not possible the figure recorded under profiles/ is quoted and labelled as such.
        traffic, traffic_source, counters = None, None, {}
        if world == 1 and not a.no_traffic:
            counters, why_not = measure_counters(a.workload, n)
            traffic, traffic_source = traffic_of(counters)
            if traffic is None:
                traffic_source = why_not
        if traffic is None:
            why = traffic_source
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    t = json.load(f).get(a.workload)
                if t and t["bytes_per_gpu"] == n:
                    traffic = t["traffic_bytes"]
                    traffic_source = "recorded in %s (not measured in this run%s)" % (t["source"], ": " + why if why else "")
            except Exception:
                pass
        # What actually bounds table-driven AES on this chip: the LDS serves 32 table lookups per
        # clock per CU.  Lookups per 16-byte block: 16 per round, minus the two rounds CTR shares between counters;
        # GCM adds 16 ds_read_b128 of GHASH per block = 32 lookup slots.  The ceiling is quoted at the clock measured
        # under this very load, so achieved / ceiling is the kernel's distance from its own bound.
        lds_ceiling = None
        lookups = {"ctr": 128, "gcm": 128 + 32, "ecb": 160, "ocb": 160, "ocb-dec": 160, "cbc-dec": 160, "cfb-dec": 160,
                   "xts": 224}.get(a.workload)               # the serial chains are latency-bound: no lookup ceiling
        if sclk_mhz and lookups:
            cus = torch.cuda.get_device_properties(0).multi_processor_count
            ceil_gbs = cus * sclk_mhz * 1e6 * 32.0 / lookups * 32.0 / 1e9
            lds_ceiling = {"lookups_per_block": lookups, "lookups_per_clk_per_cu": 32, "cus": cus,
                           "sclk_mhz_under_load": round(sclk_mhz, 0), "ceiling_gbs": round(ceil_gbs, 1),
                           "frac_of_ceiling": round(achieved / ceil_gbs, 4)}
        # Both pipes, in shader clocks per 16-byte block per CU: the LDS serves 32 lookups per
        # clock, and every VALU instruction of a stream that contains v_perm_b32 costs 4.43 SIMD-cycles
        # (profiles/r04_issuebench.log); VALU instructions per block = SQ_INSTS_VALU x 64 / blocks, from the PMC pass of
        # THIS run (or, where no pass could run, the recorded figure, labelled).  The two do not overlap freely (they
        # share the register-file ports): the dependency-free ceiling of the CTR mix is 4.56 clk per block, not max(valu, lds).
        pipes = None
        valu_per_block, valu_source = valu_per_block_of(counters, n, a.workload)
        if sclk_mhz and lookups and valu_per_block and a.workload in ("ctr", "gcm", "xts") and n >= (256 << 20):
            cus = torch.cuda.get_device_properties(0).multi_processor_count
            blocks = n / 16.0
            pipes = {"achieved_clk_per_block_per_cu": round(cus * sclk_mhz * 1e6 * kern_ms * 1e-3 / blocks, 3),
                     "lds_clk_per_block_per_cu": round(lookups * 2.08 / 64.0, 3),
                     "valu_clk_per_block_per_cu": round(valu_per_block * 4.43 / 256.0, 3),
                     "valu_insts_per_block": valu_per_block,
                     "valu_insts_source": valu_source,
                     "source": "4.43 cycles per VALU instruction and 2.08 clk per ds_read_b32 from profiles/r04_issuebench.log; "
                               "clock measured in this run"}
        names = {"ctr": "AES-128-CTR", "ecb": "AES-128-ECB", "xts": "AES-256-XTS 4 KiB sectors", "gcm": "AES-128-GCM",
                 "cbc-dec": "AES-128-CBC decrypt", "cfb-dec": "AES-128-CFB decrypt",
                 "cbc-enc": "AES-128-CBC encrypt, ONE serial chain", "cmac": "AES-128-CMAC, ONE serial chain",
                 "ocb": "AES-128-OCB", "ocb-dec": "AES-128-OCB decrypt"}
        line = {
            "metric": "GiB/s encrypted (AES-128-CTR, 1 GiB buffer per GPU)" if a.workload == "ctr"
                      else "GiB/s encrypted (%s)" % names[a.workload],
            "value": round(total_gib / elapsed, 2), "unit": "GiB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "warmup_extra_steps": extra,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s, %d MiB per GPU, device-resident, splitmix64 seed %d"
                                   % (names[a.workload], n >> 20, seed),
                       "parallelism": "shard%d" % world, "bytes_per_gpu": n},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes": int(algo_bytes),
                         "kernel_ms": round(kern_ms, 4),
                         "kernel_ms_min": round(min(per_step_ms), 4),
                         "measured_copy_ceiling_gbs": None if copy_gbs is None else round(copy_gbs, 1),
                         "lds_ceiling": lds_ceiling, "pipes": pipes},
            "verified": verify,
        }
        if world > 1:
            # roofline above = rank 0's kernel; here every GPU's, and the node aggregate against N x peak
            line["roofline"]["per_gpu_achieved"] = [round(x, 1) for x in per_gpu]
            line["roofline"]["per_gpu_frac"] = [round(x / HBM_PEAK_GBS, 4) for x in per_gpu]
            line["roofline"]["aggregate"] = {"achieved": round(sum(per_gpu), 1), "peak": HBM_PEAK_GBS * world,
                                             "frac": round(sum(per_gpu) / (HBM_PEAK_GBS * world), 4)}
            line["verified_shards"] = "every rank hashed its whole shard against the reference's digest" \
                if (a.workload == "ctr" and n == GIB and not a.no_verify) else "head of every shard against the oracle"
        if sustained:
            # whole-job sustained rate = the sum of the ranks' own back-to-back rates (no collective inside the run)
            line["sustained"] = {"value": round(sum(sus_all), 2), "unit": "GiB/s", "seconds": round(sustained["seconds"], 3),
                                 "steps": sustained["steps"], "ms_per_step": round(sustained["ms_per_step"], 4),
                                 "sclk_mhz": None if sclk_mhz is None else round(sclk_mhz, 0),
                                 "note": "the same step back to back after the K timed steps; not part of `value`"}
            line["roofline"]["frac_sustained"] = round(2.0 * n / (sustained["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        line.update(coll_info)
        line.update(gather_info)
        if c_gather is not None:
            line["c_gather"] = c_gather
        if world == 1 and a.workload == "ctr" and n == GIB and not a.no_other_configs:
            line["other_configs"] = other_configs(torch, uaes, dev, st)
        if not a.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(a.workload)
        print(json.dumps(line), file=_LINE_OUT, flush=True)
    if stuck_extra and not dist:
        sys.stdout.flush()
        os._exit(0)                             # a thread still inside the C host's gather: do not wait for it
    if dist:
        sys.stdout.flush()
        if coll_info.get("rccl_stuck_thread") or stuck_extra:
            os._exit(0)                         # a thread still inside RCCL: tearing the groups down could hang with it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
