"""BASELINE.json configurations at their FULL sizes under `pytest -m gpu` (one MI355X):

  C3  AES-256-XTS over 2^20 sectors of 4 KiB: SHA-256 of all 4 GiB of ciphertext against the
      digest the compiled reference produced (tests/golden/digests.json);
  C5  the 8 GiB AES-128-CTR stream as its eight 1 GiB shards, each with the counter offset
      g * 2^26 its GPU would use (uaes_ctr_xcrypt_at, incBlock's 56-bit add micro_aes.c:421-427),
      processed one after another on this GPU: SHA-256 of the concatenation against the
      reference's digest of the whole stream -- everything C5 needs except eight physical GPUs;
  and bench.py's N-rank code path, dry-run with two gloo ranks sharing the one GPU.

Inputs are generated on the GPU (bench.splitmix_device, the SURVEY 8d stream)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest

import micro_aes_amd as uaes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GIB = 1 << 30


def _digests(golden_dir):
    with open(os.path.join(golden_dir, "digests.json")) as f:
        return json.load(f)


def _update(h, t):
    step = 1 << 28
    for o in range(0, t.numel(), step):
        h.update(t[o:o + step].cpu().numpy().tobytes())


def test_C3_xts256_all_2p20_sectors(golden_dir):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    nsec = 1 << 20
    src = bench.splitmix_device(torch, 3, nsec * 4096, 0, dev)
    dst = torch.empty_like(src)
    uaes.xts_sectors_dev(bytes(range(64)), 0, 4096, nsec, src, dst, encrypt=True)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    _update(h, dst)
    assert h.hexdigest() == _digests(golden_dir)["C3_xts256_2p20_sectors_seed3"]["sha256"]
    # and back, in place
    uaes.xts_sectors_dev(bytes(range(64)), 0, 4096, nsec, dst, dst, encrypt=False)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)


def test_C5_ctr128_8GiB_as_eight_shards(golden_dir):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    key, ctr0 = bytes(range(16)), bytes(range(0xF0, 0xFC)) + b"\0\0\0\1"
    h = hashlib.sha256()
    dst = torch.empty(GIB, dtype=torch.uint8, device=dev)
    for g in range(8):
        # GPU g owns bytes [g * 2^30, (g+1) * 2^30) of the stream and counter offset g * 2^26
        src = bench.splitmix_device(torch, 2, GIB, g * (GIB // 8), dev)
        uaes.ctr_xcrypt_dev(key, ctr0, g * (GIB // 16), src, dst)
        torch.cuda.synchronize()
        # each shard against the reference's own digest of it (PRESET_COUNTER build, make_fixtures.py --big;
        # shard 0 is C2) -- what every rank of `bench.py --gpus N` checks
        hs = hashlib.sha256()
        _update(hs, dst)
        assert hs.hexdigest() == _digests(golden_dir)["C5_shard_%d" % g]["sha256"], g
        if g == 0:
            assert hs.hexdigest() == _digests(golden_dir)["C2_ctr128_1GiB_seed2"]["sha256"]
        _update(h, dst)
        del src
    assert h.hexdigest() == _digests(golden_dir)["C5_ctr128_8GiB_seed2"]["sha256"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("workload", ["ctr", "xts"])
def test_bench_two_rank_dry_run(workload):
    """bench.py --gpus 2 as the driver launches it, but gloo + --single-device so that both
    ranks share cuda:0: shard offsets, barrier-bracketed timing, MAX over ranks, per-rank
    verification against the oracle, one JSON line from rank 0."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--bytes", str(64 << 20), "--settle-ms", "0", "--sustain-s", "0.2",
           "--workload", workload]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["verified"] is True
    assert line["value"] > 0 and line["config"]["parallelism"] == "shard2"
    assert "cpu_baseline" not in line            # rank 0 at N=1 only


def _one_json_line(r):
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    # ONE JSON line and nothing else on stdout (RCCL's version banner once landed behind it: bench.py main())
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launch_verifies_every_shard():
    """`python bench.py --gpus 2` with NO launcher -- the shape of the driver's N=1 command: bench.py starts
    its own two ranks (torch.distributed.run, free port).  Dry run on this box's one GPU (gloo,
    --single-device) at the full 1 GiB per rank, so that EVERY rank hashes its whole shard against the
    reference's digest of that shard of the C5 stream (PRESET_COUNTER build of the reference)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--settle-ms", "0", "--sustain-s", "0.5"]
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert line["verified_shards"].startswith("every rank hashed its whole shard")
    assert len(line["roofline"]["per_gpu_achieved"]) == 2
    agg = line["roofline"]["aggregate"]
    assert agg["peak"] == 16000.0 and abs(agg["achieved"] - sum(line["roofline"]["per_gpu_achieved"])) < 1.0


def test_bench_line_measures_its_own_traffic_and_clock():
    """the N = 1 line as the driver reads it: `roofline.traffic` from the two rocprofv3 PMC passes the run makes over a
    child of itself -- the algorithmic bytes plus no more than a few per cent (it once counted the clock sample's
    extra steps) -- and `roofline.lds_ceiling` from the shader clock sampled under the load"""
    import shutil
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu"]
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    roof = line["roofline"]
    assert line["n_gpus"] == 1 and line["verified"] is True and roof["algorithmic_bytes"] == 2 << 30
    assert 0.3 < roof["frac"] < 0.6 and abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
        assert roof["traffic_source"].startswith("measured in this run"), roof["traffic_source"]
        assert 1.0 <= roof["traffic"] / roof["algorithmic_bytes"] < 1.05, roof
    lc = roof["lds_ceiling"]
    assert lc and lc["lookups_per_block"] == 128 and 1500 < lc["sclk_mhz_under_load"] < 2500
    assert 0.6 < lc["frac_of_ceiling"] < 1.0, lc
    pp = roof["pipes"]
    assert 4.0 < pp["lds_clk_per_block_per_cu"] < 4.3
    if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
        # measured by the SQ_INSTS_VALU pass of this very run (213 for the round-4 kernel), not a copied constant
        assert pp["valu_insts_source"].startswith("measured in this run") and 190 < pp["valu_insts_per_block"] < 225, pp
    assert pp["lds_clk_per_block_per_cu"] < pp["achieved_clk_per_block_per_cu"] < 5.2, pp
    # the sustained rate (>= 2 s of the same step back to back) rides next to `value`; the 12 ms window
    # behind `value` may catch the clocks a little high, never the other way round by more than noise
    # configs[3] (GCM, 1 GiB) and configs[2] (XTS-256, 2^20 x 4 KiB) ride in the driver's line,
    # each timed after the headline and checked against the compiled reference's tag / digest
    oc = line["other_configs"]
    assert oc["gcm_c4"]["verified"] is True and oc["xts_c3"]["verified"] is True, oc
    assert oc["gcm_c4"]["bytes"] == GIB and oc["xts_c3"]["bytes"] == 4 * GIB
    assert 0.25 < oc["gcm_c4"]["frac"] < 0.45 and 0.2 < oc["xts_c3"]["frac"] < 0.35, oc
    # ... and the C host's gather went through RCCL (ncclSend / ncclRecv to itself) with the C2 digest at the end
    cg = line["c_gather"]
    assert cg["forced_self_send"] is True and cg["rccl_sends"] == 2 and cg.get("stream_digest_ok") is True, cg
    sus = line["sustained"]
    assert sus["seconds"] >= 1.9 and sus["steps"] >= 1000 and sus["unit"] == "GiB/s"
    assert sus["value"] <= line["value"] * 1.04, (sus, line["value"])      # (measured: 0.99 .. 1.014 of the window's rate)
    assert sus["value"] >= line["value"] * 0.85, (sus, line["value"])
    assert abs(roof["frac_sustained"] - 2.0 * (1 << 30) / (sus["ms_per_step"] * 1e-3) / 1e9 / roof["peak"]) < 1e-3
    assert sus["sclk_mhz"] == lc["sclk_mhz_under_load"]


def test_bench_two_ranks_rccl():
    """the real thing on a multi-GPU box: two ranks, two devices, RCCL (nccl backend with device_id), the
    ciphertext all-gather included"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs: this box has %d" % torch.cuda.device_count())
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--gather"]
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert line["n_gpus"] == 2 and line["verified"] is True and line["gather_ms"] > 0


def test_bench_one_rank_force_collective_runs_rccl_on_one_gpu():
    """`bench.py --gpus 1 --force-collective` brings up a ONE-rank RCCL group (torch's "nccl"
    backend), runs the barrier / reductions / ciphertext all-gather through it and the C host's gather through ncclSend /
    ncclRecv to itself: on a one-GPU box this is the first and only execution of the code the 8-GPU run depends on.
    The gathered stream (1 GiB, = C2) must match the reference's digest through BOTH gathers."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "UAES_GATHER_FORCE_RCCL", "UAES_GATHER_FAIL_SEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--force-collective",
           "--sustain-s", "0.2", "--no-traffic", "--no-cpu"]
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert line["n_gpus"] == 1 and line["verified"] is True and line["value"] > 0
    assert line["collective_backend"] == "nccl (RCCL)", line["collective_backend"]
    assert line["rccl_ranks"] == 1
    assert line["gather_ms"] > 0 and line["gather_backend"].startswith("RCCL all_gather_into_tensor, ONE rank")
    assert line["gathered_stream_digest_ok"] is True
    cg = line["c_gather"]
    assert cg.get("stream_digest_ok") is True and cg["ms"] > 0, cg
    assert cg["forced_self_send"] is True and cg["rccl_sends"] == 2 and cg["rccl_recvs"] == 2 and cg["rccl_comm_inits"] == 1, cg


def test_bench_two_ranks_survive_an_rccl_failure():
    """`bench.py --gpus 2` with the default nccl backend, both ranks on this box's one GPU and the RCCL
    bring-up forced to fail on every rank (UAES_BENCH_FORCE_NCCL_FAIL): the run must fall back to gloo for the control
    plane, still time and verify both shards, and say what happened in `collective_backend`."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(UAES_BENCH_FORCE_NCCL_FAIL="1", UAES_BENCH_RCCL_WAIT_S="30")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--single-device", "--bytes", str(64 << 20), "--settle-ms", "0", "--sustain-s", "0.2"]      # gather + C gather: defaults
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert line["n_gpus"] == 2 and line["verified"] is True and line["value"] > 0
    assert line["collective_backend"].startswith("gloo (nccl init failed: RuntimeError: forced by"), line["collective_backend"]
    # without RCCL the ciphertext gather goes to rank 0 over gloo (the dry run's stand-in) and says so; the
    # concatenated stream is checked shard by shard
    assert line["gather_ms"] > 0 and line["gather_backend"].startswith("gloo gather to rank 0")
    assert line["encrypt_plus_gather_gib_s"] > 0 and line["gathered_stream_digest_ok"] is True
    # the C host's gather needs no RCCL when every shard sits on the root's device: it ran, and shard 0 is the step's output
    assert line["c_gather"].get("shard0_equals_own_step") is True, line["c_gather"]
    assert line["sustained"]["value"] > 0


def test_bench_two_ranks_real_rccl_bring_up_on_one_device():
    """the same WITHOUT forcing: two ranks really ask RCCL for a communicator on the same device.  Whatever RCCL does
    with that (it refuses duplicate devices), the bench line must come out verified, on RCCL or on the gloo fallback."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(UAES_BENCH_RCCL_WAIT_S="60")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--single-device", "--bytes", str(64 << 20), "--settle-ms", "0", "--sustain-s", "0.2"]
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert line["collective_backend"].startswith(("gloo (nccl init failed", "nccl (RCCL)")), line["collective_backend"]


def test_bench_eight_ranks_flagless_dry_run_reports_everything_8e_lists():
    """the driver's multi-GPU command carries no flag but --gpus/--steps/--warmup.  Eight ranks on this
    box's one GPU (--single-device is the only addition): the line must hold the gather, encrypt + gather, the digest of
    the concatenated 8 GiB stream against the reference's C5 digest, and the C host's own gather with the same digest --
    RCCL refuses eight ranks on one device, so the collective is the labelled gloo stand-in here."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(UAES_BENCH_RCCL_WAIT_S="60")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--single-device"]
    line = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200))
    assert line["n_gpus"] == 8 and line["verified"] is True and line["config"]["bytes_per_gpu"] == GIB
    assert line["gather_ms"] > 0 and line["encrypt_plus_gather_gib_s"] > 0
    assert line["gathered_stream_digest_ok"] is True and "whole 8 GiB stream" in line["gathered_stream_check"]
    cg = line["c_gather"]
    assert cg.get("stream_digest_ok") is True and cg["shard0_equals_own_step"] is True and cg["ms"] > 0, cg
